"""bench.py -- headline benchmark of the Line3D++ matching/scoring hot path on MI355X.

A "step" = one full pass of the hot path over one synthetic scene:
    Line3D::matchImages (pair matching -> orientation filter -> 3D scoring -> inverse matches ->
    hypothesis collapse) + the affinity fill of Line3D::reconstruct3Dlines.
Metric (BASELINE.json): million segment-pair scores per second = sum over matched directed view pairs
of Ms*Mt, divided by the wall time of the step, whole job over all ranks.  Segment arrays are resident
in HBM before the timed region (uploaded at addImage).  Workload at N=1: BASELINE config C1
(synthetic 64 views x 2000 segments/view, 10 visual neighbours).

N>1 (one rank per GPU over RCCL; `python bench.py --gpus N` without WORLD_SIZE starts the N ranks itself through
torch.distributed.run, under torchrun it is a rank): the halo form of line3dpp_amd/dist.py -- views cut into contiguous
ranges, a rank matches the pairs of its views, the pairs across a cut travel point to point in compact form while the
rest is being matched, the list pass, the tail of phase B and the affinity fill are sharded by the same views, their
records / outputs / similarities are exchanged in place over xGMI and every rank ends with the complete result.
"scaling": "weak" (default for N > 1; SURVEY.md 8e: pairs are independent units): the scene of N ranks is N neighbourhood
rings of the configuration's size in one scene (N x 64 views x 2000 segments for C1: per-GPU work fixed, at N = 1 exactly
the configuration); `--scaling strong` runs the SAME scene at every N.  No multi-GPU box has been available to the
builder: at N = 1 the line carries `multi_gpu_model`, the expected N-GPU time term by term from this run's phase times.

Prints ONE JSON line with
  `roofline`      the pair-matching kernel, timed with HIP events on its launch stream; VALU-issue roof priced with the
                  instruction mix of the kernel (PMC, keyed by build id) and the per-class issue cycles measured on this
                  kind of box (profiles/*_valu_calibration.json); `useful_frac`; the HBM figure beside it
  `cpu_baseline`  the reference's own OpenMP CPU path (oracle/_ref), timed on this box, rank 0, N=1
  `phase_ms`      GPU time of begin / phase A / phase B / affinity, from a few separate UNTIMED steps with all ten HIP
                  events of a call on: the timed steps record only the pair around the match kernel (an event between
                  two kernels is a ~6 us gap on the stream; l3d_set_timing_level)
  `parity`        the HIP result of the benchmarked scene against that very reference run -- or, with --parity-digest,
                  against the stored record of a reference run of the FULL configuration (tests/full_digest.py)
  `cold_ms` / `second_scene_ms`   the first matchImages + affinity of a fresh context (allocations, pool growth and
                  extra chain rounds included) and of a fresh context for ANOTHER scene of the same size afterwards;
                  `cold.fresh_process`: the first call of a fresh PROCESS (subprocess), beside its context creation
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def reference_oracle():
    """(constructor, kind) of the reference's own code; None when oracle/_ref is absent -- the caller reports that as a
    FAILED parity check: there is no downgrade to the restatement"""
    from oracle import oracle as O
    if not O.have_reference():
        return None, None
    kind = "release" if O.have_release() else True
    return (lambda threads: O.Oracle(threads=threads, reference=kind)), kind


def run_cpu_baseline(scene, args, pair_tests, kNN, l3d):
    """The reference's own OpenMP CPU path (oracle/_ref: line3D.cc / view.cc compiled in place, Release flags) on this
    box's host cores, and -- because that run IS the reference result for the benchmarked scene -- the parity check of
    the HIP result against it.  The reference's structure (std::list / std::map / priority_queue per row) stops scaling
    long before 256 threads, so the thread count is picked by a short scan on a sub-scene.  The sample is the full
    workload while it stays below ~6e9 pair tests (C0, C1, C3: 5-25 s) or when --full-parity asks for it (C2: minutes);
    otherwise a slice of consecutive views at the configured size and neighbour count.  Returns (cpu_baseline, parity)."""
    from tests import helpers as H
    Oracle, use_ref = reference_oracle()
    if Oracle is None:
        why = "oracle/_ref/libl3d_ref.so is missing (built by oracle/Makefile where /root/reference exists): no reference, " \
              "no parity claim"
        return None, {"config": args.config, "checked": False, "ok": False, "error": why}
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if args.cpu_threads:
        best_t, scan = args.cpu_threads, {}
    else:
        sub = H.ring_slice(scene, 0, min(8, scene.n_views))
        sub_tests = sub.pair_tests()[0]
        scan = {}
        for t in sorted({1, 8, 16, 32, 64, ncpu}):
            if t > ncpu:
                continue
            o = Oracle(t); o.add_scene(sub)
            t1 = time.perf_counter(); o.match_images(kNN=kNN); o.compute_affinity()
            scan[t] = round(sub_tests / (time.perf_counter() - t1) / 1e6, 1)
            if scan[t] < 0.5 * max(scan.values()):
                break                                    # past the knee: more threads only get slower
        best_t = max(scan, key=scan.get)
    sample, sample_tests, what = scene, pair_tests, f"full {args.config} workload once"
    if pair_tests > 6e9 and not args.full_parity:
        n = scene.n_views
        while n > 4 and H.ring_slice(scene, 0, n).pair_tests()[0] > 5e9:
            n -= 1
        sample = H.ring_slice(scene, 0, n)
        sample_tests = sample.pair_tests()[0]
        what = f"slice of the first {n} views of {args.config} (configured segments/view and neighbour count) once"
    o = Oracle(best_t); o.add_scene(sample)
    t1 = time.perf_counter()
    o.match_images(kNN=kNN); o.compute_affinity()
    cdt = time.perf_counter() - t1
    cpu = {"value": round(sample_tests / cdt / 1e6, 2), "unit": "M segment-pair scores/s", "cores": best_t,
           "kind": "reference", "host_cpus": ncpu, "thread_scan_M_per_s": scan,
           "sample": f"{what} ({sample_tests} pair tests, {cdt:.2f} s) with the thread count that scored best on an "
                     f"8-view sub-scene; the reference's own OpenMP CPU path (line3D.cc/view.cc compiled in place, oracle/_ref"
                     + (", -O3 -DNDEBUG)" if use_ref == "release" else ", -O2)")}
    # ---- parity of the HIP result with that reference run (same scene, same parameters) ----
    if sample is scene:
        g = l3d                                          # the context of the timed steps holds the last step's result
    else:
        from line3dpp_amd.api import Line3D
        g = Line3D(device=torch.cuda.current_device()); g.add_scene(sample)
        assert g.matchImages(kNN=kNN) and g.computeAffinity()
    d = H.full_result_diff(g, o, sample)
    parity = {"config": args.config, "checked": True, "against": "reference (live run of oracle/_ref)", "scene": what,
              "ok": d["ok"],
              "surviving_matches": d["surviving"], "set_diff": d["set_diff"], "best_hypotheses": d["best"],
              "best_set_diff": d["best_set_diff"], "best_choice_diff": d.get("best_choice_diff"),
              "affinity_entries": d["affinity_entries"], "affinity_set_diff": d["affinity_set_diff"],
              "inexact_overlap_or_depth_fields": d["inexact_phase_a_fields"],
              "max_rel": max(d["max_rel_score3D"], d.get("max_rel_endpoints") or 0.0, d["max_rel_affinity"] or 0.0),
              "max_rel_score3D": d["max_rel_score3D"], "max_rel_endpoints": d.get("max_rel_endpoints"),
              "max_rel_affinity": d["max_rel_affinity"], "order_rows": d["order_rows"], "tie_rows": d["tie_rows"],
              "tolerance": H.REL_TOL}
    return cpu, parity


def digest_parity(scene, args, l3d):
    """The HIP result of the FULL benchmarked scene against the stored record of a run of the reference's own code on
    it (tests/full_digest.py, written by tools/ref_digest.py): identical sets / order / phase-A fields through SHA-256
    digests, float fields at REL_TOL against the full arrays beside oracle/_ref or, where those did not travel, their
    committed strided sample (floats_checked: "full" | "sample"); with neither the block says ok: false."""
    from tests import full_digest as FD
    from tests import helpers as H
    meta, floats = FD.load_reference(args.config)
    if meta is None:
        return {"config": args.config, "checked": False, "ok": False,
                "error": f"tests/golden/full/{args.config}.json is missing (tools/ref_digest.py {args.config})"}
    if meta["scene_sha256"] != FD.scene_hash(scene):
        return {"config": args.config, "checked": False, "ok": False, "error": "stored reference record is of another scene"}
    if floats is None:   # neither the full float arrays nor their committed sample: no half-checked "ok"
        return {"config": args.config, "checked": False, "ok": False,
                "error": f"no float arrays for the stored record (oracle/_ref/cache/full_{args.config}.npz, "
                         f"tests/golden/full/{args.config}_floats_sample.npz)"}
    exact, fl = FD.result_record(l3d, scene, False)
    r = FD.compare(exact, fl, meta["exact"], floats, H.REL_TOL)
    r["ok"] = bool(r["ok"] and r["floats_checked"] and r["max_rel"] is not None)
    return {"config": args.config, "checked": True,
            "against": f"stored record of the reference's own code ({meta['reference_library']}, objects md5 {meta.get('reference_objects_md5')}, "
                       f"{meta['seconds']} s on {meta['threads']} threads, tools/ref_digest.py)",
            "scene": f"full {args.config} workload", "ok": r["ok"], "surviving_matches": r["surviving"],
            "set_diff": 0 if not r["differing_views"] else None, "differing_views": r["differing_views"][:16],
            "best_hypotheses": r["best"], "best_equal": r["best_equal"], "affinity_entries": r["affinity_entries"],
            "affinity_pairs_equal": r["affinity_pairs_equal"], "floats_checked": r["floats_checked"],
            "max_rel": r["max_rel"], "max_rel_by_field": r.get("max_rel_by_field"), "tolerance": H.REL_TOL}


# ---- roofline of k_match_pairs --------------------------------------------------------------------------------------
# class of the SQ_INSTS_VALU_* counters -> the calibrated stream that prices it (tools/valu_calib.hip)
_CLASS_OPS = {"ADD_F32": ["add_f32"], "MUL_F32": ["mul_f32"], "FMA_F32": ["fma_f32"], "TRANS_F32": ["rcp_f32", "rsq_f32", "sqrt_f32"],
              "ADD_F64": ["add_f64"], "MUL_F64": ["mul_f64"], "FMA_F64": ["fma_f64", "fmac_f64"], "TRANS_F64": ["rcp_f64", "rsq_f64"],
              "CVT": ["cvt_f64_f32", "cvt_f32_f64"], "INT32": ["add_u32", "and_b32", "lshl_b32", "lshl_add_u32"], "INT64": ["lshl_b64"],
              # everything the class counters do not cover (compares, selects, moves, min/max, lane ops, ldexp, the
              # division helpers): the streams measured for such instructions (without VCC hazards); the roof is priced
              # with the CHEAPEST of them -- the choice that makes the ceiling highest, i.e. the fraction smallest
              "other": ["mov_b32", "mov_b64", "cndmask_s", "cmp_f32_s", "cmp_f64_s", "min_f32", "max_f32", "med3_f32", "max_f64", "readlane",
                        "writelane", "ldexp_f64", "cmp_class_f64", "div_scale_f64", "div_fmas_f64", "div_fixup_f64"]}


def valu_roof(pmc, calib):
    """mix-weighted VALU issue ceiling: (G wave64 instructions/s at 1024 SIMDs x 2.4 GHz, basis dict) or (None, why)"""
    cls = pmc.get("valu_class_insts_per_launch")
    if not cls or not calib:
        return None, "needs the per-class instruction counters (tools/pmc_bench.sh) and profiles/*_valu_calibration.json"
    ops = {o["op"]: o for o in calib["ops"]}
    total = pmc["valu_insts_per_launch"]
    counted = {k: v for k, v in cls.items() if k in _CLASS_OPS and k != "other"}
    mix = dict(counted, other=max(total - sum(counted.values()), 0))
    cyc = {}
    for k in mix:
        have = [ops[n]["cycles_per_unit_simd_best"] for n in _CLASS_OPS[k] if n in ops]
        if not have:
            return None, f"calibration lacks a stream for class {k}"
        cyc[k] = min(have) if k == "other" else sum(have) / len(have)
    mean_cycles = sum(mix[k] * cyc[k] for k in mix) / max(total, 1)
    peak = 1024 * 2.4 / mean_cycles
    other_all = [ops[n]["cycles_per_unit_simd_best"] for n in _CLASS_OPS["other"] if n in ops]
    mean_if_other_avg = mean_cycles + mix["other"] * (sum(other_all) / len(other_all) - cyc["other"]) / max(total, 1)
    return peak, {"simds": 1024, "clock_ghz": 2.4, "mean_issue_cycles_per_instruction": round(mean_cycles, 3),
                  "peak_if_other_priced_at_the_mean_of_its_streams": round(1024 * 2.4 / mean_if_other_avg, 1),
                  "cycles_per_class": {k: round(v, 3) for k, v in cyc.items()},
                  "mix_fraction": {k: round(v / max(total, 1), 4) for k, v in mix.items()},
                  "calibration": calib.get("_file"), "note": "issue cycles per wave64 instruction and SIMD measured with "
                  "single-instruction streams (tools/valu_calib.hip: all instructions of a launch / (kernel time x measured clock x 1024 "
                  "SIMDs)); `other` priced at the cheapest of its streams"}


def load_json_newest(pattern, pred=lambda d: True):
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if pred(d):
            d["_file"] = os.path.relpath(path, ROOT)
            return d
    return None


def multi_gpu_model(pairs, M, kNN, phase, lists_ms, record_bytes, tail_bytes=0.0, sim_bytes=0.0):
    """What the halo form of the multi-GPU call (line3dpp_amd/dist.py) is expected to take on N GPUs of one node, term by
    term, from THIS run's single-GPU phase times and the plan the N ranks would follow (l3d_plan_shards): no multi-GPU
    box has been available to measure it, so the terms are printed for whoever has one.  Two readings per N:
    `strong` -- this very scene cut into N view ranges; `weak` -- N rings of this scene's size in one scene (what
    `bench.py --gpus N` runs by default), per-rank work as on one GPU plus what grows with N (the records, outputs and
    similarities of N - 1 peers arrive; the chain of inverse matches runs over the records of all ranks).
    Assumptions stated in the output: 153 GB/s per xGMI link and direction, direct exchanges (one part per link of the full
    xGMI mesh), expansion of a received slot at the round-2 measured 2.8e-5 us, 0.03 ms of host latency per
    synchronisation point, 0.02 ms per status / count exchange; of the tail of phase B (finish - list pass) the part that
    is per view (scores, filterMatches, outputs, medians: divided by N since round 5, l3d_tail_shard_*) is taken as 0.55,
    the rest (the chain: a fixed point over all records) as replicated -- the split of profiles/r04_final_kernel_stats."""
    from line3dpp_amd import dist as l3d_dist
    cost = np.asarray([M[s] * M[t] for s, t in pairs], np.float64)
    per_view = 0.55
    out = {"MODELLED_NOT_MEASURED": "no multi-GPU node has been available: every figure below is arithmetic on this run's single-GPU phase times",
           "assumptions": {"link_GB_per_s": 153.0, "exchanges": "direct: one part per link, all peers at once (full xGMI mesh); a ring would take (N-1) part times",
                            "expand_us_per_slot": 2.8e-5, "host_sync_ms": 0.03, "status_or_count_exchange_ms": 0.02,
                            "host_sync_points": "round 6: 4 (list pass + the one status exchange in front of the record gather, tail count, "
                                                "commit, affinity finish) + 1 exchange that carries the tail's counts AND status (round 5: 4 + 4 + 1); "
                                                "the sharded entries do not wait for the device when RCCL orders itself behind the stream (l3d_shard_options)",
                            "records": "a rank receives the record slabs of the ranks its views depend on (dist.shard_needs): all lower ranks for a "
                                       "ring cut into view ranges (strong), NONE for N rings without pairs between them (weak) -- and its chain "
                                       "covers those records only",
                            "tail_per_view_fraction": per_view, "record_bytes": int(record_bytes), "tail_output_bytes": int(tail_bytes),
                            "similarity_bytes": int(sim_bytes)}}
    tail_ms = max(phase["finish"] - lists_ms, 0.0)
    t1 = phase["begin"] + phase["match"] + phase["finish"] + phase["affinity"]
    syncs = 4 * 0.03 + 2 * 0.02
    for n in (2, 4, 8):
        plan = l3d_dist.plan_halo(pairs, M, n)
        pb = plan["pair_bounds"].astype(np.int64)
        share = max(cost[pb[r]:pb[r + 1]].sum() for r in range(n)) / cost.sum()
        halo_in = [0] * n
        for r in range(n):
            for (q, f, k) in plan["runs"][r]:
                halo_in[q] += sum(M[pairs[p][0]] * kNN for p in range(f, f + k))
        halo_slots = max(halo_in)
        link = 153e9
        terms = {"begin": phase["begin"], "match_own_pairs": phase["match"] * share,
                 "halo_exchange_hidden_behind_matching_MB": round(4e-6 * halo_slots, 2),
                 "expand_received_pairs": 2.8e-5 * 1e-3 * halo_slots, "list_pass_own_views": lists_ms / n,
                 "gather_records_direct": 1e3 * record_bytes / n / link, "gather_records_if_ring": 1e3 * (n - 1) / n * record_bytes / link,
                 "tail_chain_replicated": tail_ms * (1 - per_view), "tail_own_views": tail_ms * per_view / n,
                 "exchange_tail_outputs": 1e3 * tail_bytes / n / link,
                 "affinity_similarities_own_views": 0.6 * phase["affinity"] / n, "affinity_bookkeeping_replicated": 0.4 * phase["affinity"],
                 "exchange_similarities": 1e3 * sim_bytes / n / link, "host_syncs": syncs}
        total = sum(v for k, v in terms.items() if not k.endswith("_MB") and not k.endswith("_if_ring"))
        # weak: every rank does one GPU's work; what arrives from N - 1 peers and what is replicated grows with N
        weak = {"own_work_as_on_one_gpu": t1, "gather_records_direct": 0.0,          # (independent rings: no rank depends on another's records)
                "exchange_tail_outputs": 1e3 * tail_bytes / link, "exchange_similarities": 1e3 * sim_bytes / link,
                "tail_chain_over_all_ranks_records": 0.0,                             # (the chain covers the rank's own records only)
                "affinity_bookkeeping_over_all_ranks": (n - 1) * 0.4 * phase["affinity"], "host_syncs": syncs}
        wtotal = sum(weak.values())
        out[str(n)] = {"strong": {"terms_ms": {k: round(v, 4) for k, v in terms.items()}, "total_ms": round(total, 4),
                                  "speedup_over_1_gpu": round(t1 / total, 2), "largest_pair_share": round(share, 4)},
                       "weak": {"terms_ms": {k: round(v, 4) for k, v in weak.items()}, "total_ms": round(wtotal, 4),
                                "throughput_over_1_gpu": round(n * t1 / wtotal, 2), "efficiency": round(t1 / wtotal, 3)},
                       # (kept at the top level for readers of earlier rounds' lines: the strong reading)
                       "terms_ms": {k: round(v, 4) for k, v in terms.items()}, "total_ms": round(total, 4),
                       "speedup_over_1_gpu": round(t1 / total, 2), "largest_pair_share": round(share, 4)}
    return out


def phase_b_roofline(tm, lists_ms, pairs, M, kNN, build, config, world):
    """The second roofline block (round 6): the LIST PASS of phase B -- k_pair_csr, the k_lists tiers, k_cand_exact, k_edges --,
    the bandwidth-style part of the path (scoringCPU + storeInverseMatches, line3D.cc:1208-1294, 1672-1699, in sparse form).
    `achieved` = algorithmic bytes of the pass / its live HIP-event time (events around the pass at timing level 2, the
    zero-block memset included); per kernel the counted traffic and the issue counters of the PMC summary of THIS build
    (tools/pmc_phase_b.sh -> profiles/*_pmc_lists_<config>.json), when there is one.
    Algorithmic bytes, from the counts the library reports for the timed scene:
      k_pair_csr    2 B (4 B from 65 535 segments per view on) of the inverse-target stream per slot of a pair that hands matches
                    over + 4 B per inverse hypothesis written + the per-pair offsets (4 B per target segment and pair)
      k_lists       per hypothesis of a list 8 B of depths + 4 B of identity, fresh and inverse alike; 40 B per candidate written
      k_cand_exact  40 B per candidate read, 4 B written
      k_edges       40 B per candidate read; 16 B per edge, 64 B per header written"""
    n_ent, n_inv, n_cand, n_hdr, n_edge = (tm.get(k, 0) for k in ("list_entries", "list_inverse", "list_candidates", "list_headers", "support_words"))
    tgt_b = 2 if max(M.values()) < 65535 else 4
    inv_pairs = [(s_, t_) for s_, t_ in pairs if t_ > s_]
    csr = sum(tgt_b * M[s_] * kNN + 4 * (M[t_] + 1) for s_, t_ in inv_pairs) + 4 * n_inv
    algo = {"k_pair_csr": csr, "k_lists": 12 * n_ent + 40 * n_cand, "k_cand_exact": 44 * n_cand, "k_edges": 40 * n_cand + 16 * n_edge + 64 * n_hdr}
    total = sum(algo.values())
    out = {"scope": "list pass of phase B: k_pair_csr + k_lists<1|2|4> + k_cand_exact + k_edges", "bound": "hbm",
           "achieved": round(total / (lists_ms * 1e-3) / 1e9, 2) if lists_ms > 0 else None, "peak": 8000.0, "unit": "GB/s",
           "frac": round(total / (lists_ms * 1e-3) / 1e9 / 8000.0, 5) if lists_ms > 0 else None,
           "live_ms": round(lists_ms, 4), "timed_with": "HIP events around the pass on its launch stream (l3d_set_timing_level 2), separate untimed steps",
           "algorithmic_bytes": dict(algo, total=total),
           "counts": {"hypotheses_in_lists": n_ent, "of_which_inverse": n_inv, "candidate_pairs": n_cand, "supported_hypotheses": n_hdr,
                      "supporting_pairs": n_edge, "slots": tm.get("slots_lo", 0) + (tm.get("slots_hi", 0) << 32)}}
    pmc = load_json_newest("*_pmc_lists_%s.json" % config, lambda d: d.get("build_info") == build) if world == 1 else None
    if pmc:
        ks = {}
        groups = {"k_pair_csr": "k_pair_csr", "k_lists": "k_lists<", "k_cand_exact": "k_cand_exact", "k_edges": "k_edges"}
        for name, prefix in groups.items():
            sel = [v for k, v in pmc["kernels"].items() if k.startswith(prefix) and not k.startswith("k_lists_huge") and v.get("avg_us")]
            if not sel:
                continue
            us = sum(v["avg_us"] * (v.get("calls_per_step") or 1) for v in sel)
            rd = sum(v.get("fetch_bytes_x2", 0) * (v.get("calls_per_step") or 1) for v in sel)
            rd_raw = sum(v.get("fetch_bytes_raw", 0) * (v.get("calls_per_step") or 1) for v in sel)
            wr = sum(v.get("write_bytes", 0) * (v.get("calls_per_step") or 1) for v in sel)
            big = max(sel, key=lambda v: v["avg_us"])
            ks[name] = {"us_per_step": round(us, 2), "algorithmic_bytes": algo[name],
                        "achieved_GB_per_s": round(algo[name] / (us * 1e-6) / 1e9, 1), "frac_of_8_TB_per_s": round(algo[name] / (us * 1e-6) / 8e12, 4),
                        "counted_read_bytes_x2": int(rd), "counted_read_bytes_raw": int(rd_raw), "counted_write_bytes": int(wr),
                        "counted_GB_per_s_x2": round((rd + wr) / (us * 1e-6) / 1e9, 1),
                        "traffic_over_algorithmic_x2": round((rd + wr) / max(algo[name], 1), 2),
                        "traffic_over_algorithmic_raw": round((rd_raw + wr) / max(algo[name], 1), 2),
                        "valu_busy_fraction": big.get("valu_busy_fraction"), "wait_share_of_wave_life": big.get("wait_share_of_wave_life"),
                        "l2_hit_rate": big.get("l2_hit_rate")}
        out["kernels"] = ks
        out["pmc_source"] = pmc["_file"] + " (rocprofv3 --pmc passes + an undisturbed --kernel-trace run, same build id)"
        out["reading"] = ("FETCH_SIZE x 2 is the guide's gfx950 correction for wide coalesced streams and an upper bound for the 8-byte "
                          "gathers of the list pass (`raw` = as counted). k_lists is bound by VALU issue (valu_busy_fraction), not by HBM: "
                          "see DESIGN.md 5.2")
    else:
        out["pmc_source"] = "no PMC summary under profiles/ for this build and workload (tools/pmc_phase_b.sh)"
    return out


def cold_call(scene, kNN, device_index, step_fn_factory):
    """first matchImages + affinity of a fresh context: (ms, timings dict, context)"""
    from line3dpp_amd.api import Line3D
    g = Line3D(device=device_index)
    g.add_scene(scene)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_fn_factory(g)()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    tm = g.timings()
    return ms, {k: tm[k] for k in ("pool_retries", "chain_extra_rounds", "chain_sweeps")}, g


_PROCESS_COLD = """
import json, sys, time
sys.path.insert(0, %r)
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
from line3dpp_amd import _lib
sc = make_config(%r)
t0 = time.perf_counter()
_lib.load()                               # dlopen of libl3dpp_hip.so and of the HIP runtime behind it
ta = time.perf_counter()
g = Line3D(device=%d)                     # l3d_create: hipSetDevice = the runtime's own start-up, events, the start-up copies / launches
tb = time.perf_counter()
g.add_scene(sc)
t1 = time.perf_counter()
ok = g.matchImages(kNN=%d) and g.computeAffinity()
t2 = time.perf_counter()
print(json.dumps({"ok": bool(ok), "create_and_add_ms": round(1e3 * (t1 - t0), 1),
                  "of_which": {"load_library_ms": round(1e3 * (ta - t0), 1), "l3d_create_ms": round(1e3 * (tb - ta), 1),
                               "add_views_ms": round(1e3 * (t1 - tb), 1)},
                  "first_call_ms": round(1e3 * (t2 - t1), 3)}))
"""


def process_cold(config, device_index, kNN):
    """the FIRST matchImages + affinity of a fresh PROCESS (nothing of the runtime warm: what a one-scene command-line run
    pays), next to the context creation that precedes it (HIP start-up, code objects, copy paths: l3d_create)"""
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", _PROCESS_COLD % (ROOT, config, device_index, kNN)], capture_output=True,
                             text=True, timeout=600)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001 -- a diagnostic beside the benchmark, never its failure
        return {"error": repr(e)[:200]}


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE): start N ranks of this script, one per GPU, through
    torch.distributed.run on 127.0.0.1 -- the form the driver itself uses for N > 1 -- and return its exit code.
    Fails loudly when the box has fewer than N GPUs (never a silent N = 1 run)."""
    import socket
    import subprocess
    if not args.launcher_selftest:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this box "
                             f"(one rank per GPU; refusing to run fewer ranks than asked for)")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def weak_scene(config, n):
    """The scene of a weak-scaling run on n ranks: n neighbourhood rings of the configuration's size in ONE scene (camIDs
    and neighbour lists offset ring by ring, each ring its own seeded instance of the configuration: ring 0 is the
    configuration itself), so that a contiguous, cost-balanced cut into n view ranges (l3d_plan_shards) gives every rank
    one GPU's work.  n = 1: the configuration."""
    from line3dpp_amd.scene import CONFIGS, Scene, make_config
    if n == 1:
        return make_config(config)
    base = 0x4C334450 + list(CONFIGS).index(config)
    views = []
    for r in range(n):
        sub = make_config(config, seed=base + 7919 * r) if config != "C0" else make_config(config)
        off = len(views)
        for v in sub.views:
            v.cam += off
            v.neighbors = [int(x) + off for x in v.neighbors]
            views.append(v)
    return Scene(views, f"{config}x{n}rings")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C1", help="BASELINE config: C0 | C1 | C2 | C3 | C4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all host cores)")
    ap.add_argument("--full-parity", action="store_true",
                    help="run the reference on the FULL configuration (C2: ~2 min, C4: ~5 min of CPU) instead of a slice")
    ap.add_argument("--parity-digest", action="store_true",
                    help="parity of the full scene against the stored reference record (tests/golden/full) instead of a live run")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-call / second-scene measurement")
    ap.add_argument("--no-extra-lines", action="store_true",
                    help="N = 1: skip the short C2 run that feeds the model of BASELINE's 8-GPU configuration; N > 1: skip the "
                         "C2 strong-scaling line printed beside the default weak-scaling C1 line")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = N rings of the configuration's size in one scene (per-GPU work fixed, default); "
                         "strong = the same scene at every N")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="start the ranks as --gpus N would, over gloo on the CPU, report them and exit (no GPU work): the "
                         "CPU test of the launcher")
    args = ap.parse_args()

    # ---- one process per GPU.  Under torchrun / torch.distributed.run (WORLD_SIZE set) this process IS a rank; started
    # plainly with --gpus N > 1 it starts the N ranks itself -- `python bench.py --gpus 8` must never run one rank and
    # print "n_gpus": 1 (round 4 did) ----
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or under "
                         f"torch.distributed.run with --nproc-per-node equal to --gpus")
    import torch.distributed as dist
    if args.launcher_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            seen = [None] * world
            dist.all_gather_object(seen, rank)
            n_backend, backend = dist.get_world_size(), dist.get_backend()
            dist.destroy_process_group()
        else:
            seen, n_backend, backend = [0], 1, "none"
        if rank == 0:
            print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks": sorted(seen),
                              "ranks_reported_by_backend": n_backend, "backend": backend}))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    if torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU, no oversubscription)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from line3dpp_amd import dist as l3d_dist
    from line3dpp_amd.api import Line3D
    from line3dpp_amd.scene import CONFIGS, make_config, make_scene

    weak = world > 1 and args.scaling == "weak"
    scene = weak_scene(args.config, world) if weak else make_config(args.config)
    cfg = CONFIGS[args.config]
    pair_tests, pairs = scene.pair_tests()
    kNN = 10

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def stepper(ctx):
        def step():
            ok = l3d_dist.match_images_sharded(ctx, rank, world, device=device, kNN=kNN)
            assert ok, "matchImages failed"
            assert l3d_dist.compute_affinity_sharded(ctx, rank, world, device=device), "affinity failed"
        return step

    def quick_line(config, scaling, steps, warmup):
        """a plain timed line of another configuration in this process group (no roofline / CPU / cold legs): the metric, the
        phase times of a few untimed steps at timing level 2 and what multi_gpu_model needs"""
        wk = world > 1 and scaling == "weak"
        sc = weak_scene(config, world) if wk else make_config(config)
        tests, prs = sc.pair_tests()
        g = Line3D(device=local_rank); g.add_scene(sc)
        st = stepper(g)
        for _ in range(warmup):
            st()
        barrier()
        t0_ = time.perf_counter()
        for _ in range(steps):
            st()
        barrier()
        dt_ = time.perf_counter() - t0_
        if world > 1:
            t_ = torch.tensor([dt_], dtype=torch.float64, device=device)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            dt_ = float(t_.item())
        ph = dict(begin=0.0, match=0.0, finish=0.0, affinity=0.0, lists=0.0)
        g.setTimingLevel(2)
        n_ph = max(1, min(steps, 3))
        for _ in range(n_ph):
            st()
            tmq = g.timings()
            ph["begin"] += tmq["begin_ms"] / n_ph; ph["match"] += tmq["match_pairs_ms"] / n_ph
            ph["finish"] += tmq["finish_ms"] / n_ph; ph["affinity"] += tmq["affinity_ms"] / n_ph; ph["lists"] += tmq.get("lists_ms", 0.0) / n_ph
        g.setTimingLevel(1)
        Mq = {v.cam: len(v.segs) for v in sc.views}
        extra = {}
        if world == 1:
            extra = dict(n_surv=sum(int(g.matches(v.cam)[1][-1]) for v in sc.views), n_best=len(g.best()[0]), record_kbytes=tmq.get("record_kbytes", 0))
        g.close()
        c_ = CONFIGS[config]
        return {"config": {"workload": ((f"{world} rings of " if wk else "") + f"{config}: synthetic {c_['n_views']} views x {c_['n_segs']} segments/view, "
                                        f"{c_['n_neighbors']} visual neighbours, kNN=10; matchImages + affinity fill"),
                           "pair_tests_per_step": tests, "directed_pairs": len(prs)},
                "scaling": scaling if world > 1 else "weak", "n_gpus": world, "steps": steps, "warmup": warmup,
                "value": round(tests / (dt_ / steps) / 1e6, 2), "unit": "M segment-pair scores/s", "ms_per_step": round(1e3 * dt_ / steps, 4),
                "phase_ms": {k: round(v, 4) for k, v in ph.items()}}, (prs, Mq, ph, extra)

    # ---- cold call: the FIRST matchImages + affinity of a fresh context (a user of Line3D::matchImages pays this once per
    # scene).  Only the code objects are warm: a tiny scene on a throw-away context loads the kernels first.
    cold = None
    if world == 1 and not args.no_cold:
        tiny = Line3D(device=local_rank); tiny.add_scene(make_scene(4, 200, n_neighbors=2, seed=3))
        assert tiny.matchImages() and tiny.computeAffinity()
        tiny.close()
        cold_ms, cold_tm, l3d = cold_call(scene, kNN, local_rank, stepper)
        cold = {"cold_ms": round(cold_ms, 3), "cold_call": cold_tm}
        if rank == 0:
            cold["fresh_process"] = process_cold(args.config, local_rank, kNN)
    else:
        l3d = Line3D(device=local_rank)
        l3d.add_scene(scene)            # segment arrays now resident in HBM
    step = stepper(l3d)

    for _ in range(args.warmup):
        step()
    kern_ms, kern_launches, phase = 0.0, 0, dict(begin=0.0, match=0.0, finish=0.0, affinity=0.0)
    lists_ms_sum, record_kbytes = 0.0, 0
    # The timed steps run the library AS SHIPPED: its default timing level (1 since round 5) records one pair of HIP events,
    # the one around the dominant kernel (roofline.achieved is its live event time), which is all this loop reads.  An
    # event between two kernels costs a ~6 us bubble on the stream and a call has ten of them at the profiling level
    # (l3d_set_timing_level 2): the per-phase times come from separate steps, untimed, at that level.
    import ctypes as C
    from line3dpp_amd import _lib
    tm_raw = _lib.Timings()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        l3d.L.l3d_get_timings(l3d.h, C.byref(tm_raw))
        kern_ms += tm_raw.match_kernel_ms; kern_launches += tm_raw.match_kernel_launches
    barrier()
    dt = time.perf_counter() - t0
    l3d.setTimingLevel(2)
    phase_steps = max(1, min(args.steps, 5))
    for _ in range(phase_steps):
        step()
        tm = l3d.timings()
        phase["begin"] += tm["begin_ms"]; phase["match"] += tm["match_pairs_ms"]
        phase["finish"] += tm["finish_ms"]; phase["affinity"] += tm["affinity_ms"]
        lists_ms_sum += tm.get("lists_ms", 0.0); record_kbytes = tm.get("record_kbytes", 0)
    tm_last = dict(tm)
    l3d.setTimingLevel(1)
    barrier()
    # (for multi_gpu_model: what the sharded tail / affinity fill would exchange -- read before the context is closed)
    n_surv = sum(int(l3d.matches(v.cam)[1][-1]) for v in scene.views) if world == 1 and rank == 0 else 0
    n_best = len(l3d.best()[0]) if world == 1 and rank == 0 else 0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    value = pair_tests / (dt / args.steps) / 1e6

    # ---- roofline of the dominant kernel (k_match_pairs), this rank's launches ----
    M = {v.cam: len(v.segs) for v in scene.views}
    if world > 1:       # the pairs this rank matches: those of its views (halo form, l3d_plan_shards)
        pbounds = l3d_dist.plan_halo(pairs, M, world)["pair_bounds"]
        f, c = int(pbounds[rank]), int(pbounds[rank + 1] - pbounds[rank])
    else:
        f, c = 0, len(pairs)
    my_pairs = pairs[f:f + c]
    # algorithmic bytes per launch (SURVEY.md §8d / DESIGN.md): per directed pair 16*(Ms+Mt) B of segment
    # records read + (32 + 2 + 16)*kNN*Ms B written: the result slots and, since the orientation filter is fused into
    # the kernel, the 2-byte inverse target of every slot (round 4; 4 bytes for views of 65 535 segments and more;
    # rounds 1-3: a 4-byte inverse-list position, 36 B per slot; SURVEY's bound is 40*kNN*Ms)
    tgt_bytes = 2 if max(M.values()) < 65535 else 4
    # (round 6: + 16 B per slot, the two 8-byte hypothesis streams of phase B written beside the slot -- hyp_p, hyp_q)
    algo_bytes = sum(16 * (M[s] + M[t]) + (32 + tgt_bytes + 16) * kNN * M[s] for s, t in my_pairs)
    my_tests = sum(M[s] * M[t] for s, t in my_pairs)
    # (one launch per step on one GPU; a rank of the halo form matches its pairs in two or three launches: the figures
    # below are per step, i.e. over all of a step's launches of the kernel)
    avg_ms = kern_ms / max(args.steps, 1)
    # two waves share a 64-row work item while the launch has few of them (k_match.hip: match_waves_per_group); the items
    # of a pair: its rows in the padded width-class layout, cut into 64 (l3d_kernels.h: tile_src_cap)
    def items_of(ms):
        return (ms + (2 if ms // 64 < 24 else 5) * 63 + 63) // 64
    wpg = 2 if sum(items_of(M[s]) for s, _ in my_pairs) <= 16384 else 1
    # launches of up to 2 048 row-form items take the tile form (16 rows per item, one wave each; l3d_kernels.h: kMatchTileMaxItems)
    tile_env = os.environ.get("L3D_MATCH_TILE")
    tile = int(tile_env) if tile_env in ("0", "16") else (16 if sum((M[s] + 63) // 64 for s, _ in pairs) <= 2048 else 0)
    hbm_achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # What binds this kernel is VALU issue, not HBM (< 0.2 B per pair test): the primary roofline is the VALU one.
    # Its inputs are PMC counters, which cannot be read from inside this process: tools/pmc_bench.sh collects them
    # (rocprofv3 --pmc, separate passes) into profiles/*_pmc_match.json TOGETHER WITH THE BUILD ID of the library they
    # were measured on, and they are only quoted here when that id is the id of the library being timed now (and the
    # workload is the same); otherwise the block says so and falls back to the HBM figure, which is always live.
    from line3dpp_amd import _lib
    build = _lib.load().l3d_build_info().decode()
    pmc = load_json_newest("*_pmc_match.json", lambda d: d.get("build_info") == build and d.get("config") == args.config) \
        if world == 1 else None
    pmc_note = (pmc["_file"] + " (rocprofv3 --pmc, same build id)") if pmc else \
        "no PMC summary under profiles/ for this build and workload (tools/pmc_bench.sh): VALU figures not quoted"
    calib = load_json_newest("*_valu_calibration.json")
    fit = load_json_newest("*_valu_fit.json", lambda d: d.get("build_info") == build)
    hbm = {"achieved": round(hbm_achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(hbm_achieved / 8000.0, 6),
           "algorithmic_bytes": algo_bytes, "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
           "traffic_over_algorithmic": round(pmc["traffic_bytes_per_launch"] / algo_bytes, 2) if pmc else None}
    common = {"kernel": f"k_match_pairs<0,false,true,{1 if tile else wpg},true,{tile}>", "kernel_ms": round(avg_ms, 4),
              "nominal_pair_tests_per_s": round(my_tests / (avg_ms * 1e-3), 1) if avg_ms > 0 else 0.0,
              "pmc_source": pmc_note, "build": build}
    if pmc:
        valu_achieved = pmc["valu_insts_per_launch"] / (avg_ms * 1e-3) / 1e9
        peak, basis = valu_roof(pmc, calib)
        if peak is None:
            # uncalibrated fallback: the fastest rate the microarchitecture guide gives (2 cycles per wave64 fp32
            # instruction on a SIMD-32) -- an upper bound of any mix's ceiling, i.e. the smallest defensible fraction
            peak, basis = 1024 * 2.4 / 2.0, {"note": "UNCALIBRATED: " + basis + "; priced at 2 cycles per instruction "
                                             "(MI355X_MICROARCH.md: wave64 v_fma_f32 on a SIMD-32), an upper bound of the ceiling"}
        cand = pmc.get("candidates") or {}
        useful = None
        if fit and cand.get("band_pairs") and cand.get("kept_slots"):
            # useful work: a walk that visits, per ROW, only the targets whose band meets the row's own band, and exact
            # tests + epilogue work for the slots that are kept -- priced with the fitted per-unit VALU costs
            # (profiles/*_valu_fit.json) -- over the VALU instructions the launch executed
            uv = (cand["band_pairs"] / 64.0) * fit["valu_per_target_visit"] + \
                 (cand["kept_slots"] / 64.0) * (fit["valu_per_drain"] + fit["valu_per_epilogue_pass"])
            useful = {"useful_frac": round(uv / pmc["valu_insts_per_launch"], 4),
                      "useful_valu_insts": round(uv), "executed_valu_insts": pmc["valu_insts_per_launch"],
                      "band_pairs": cand["band_pairs"], "prefilter_tests": cand.get("prefilter_tests"),
                      "kept_slots": cand["kept_slots"], "exact_tests": cand.get("exact_tests"), "accepted": cand.get("accepted"),
                      "exact_tests_over_accepted": round(cand["exact_tests"] / max(cand["accepted"], 1), 3) if cand.get("exact_tests") else None,
                      "valu_per_target_visit": fit["valu_per_target_visit"], "valu_per_drain": fit["valu_per_drain"],
                      "valu_per_epilogue_pass": fit["valu_per_epilogue_pass"], "fit": fit["_file"]}
        # the scalar unit: its instructions issue beside the vector ones (another wave's), one per ~4.3 cycles and SIMD
        # (calibration streams s_add_u32 / s_and_b32 / s_lshl_b32) -- in the kernel's walk it is the busier port
        scalar = None
        n_salu = (pmc.get("counters_per_launch") or {}).get("SQ_INSTS_SALU")
        if n_salu and calib:
            sc = [o["cycles_per_unit_simd_best"] for o in calib["ops"] if o.get("class") == "SALU" and o.get("cycles_per_unit_simd_best")]
            if sc:
                cyc_s = sum(sc) / len(sc)
                speak = 1024 * 2.4 / cyc_s
                sach = n_salu / (avg_ms * 1e-3) / 1e9
                scalar = {"achieved": round(sach, 2), "peak": round(speak, 1), "unit": "G SALU instr/s", "frac": round(sach / speak, 4),
                          "issue_cycles_per_instruction": round(cyc_s, 3), "salu_per_valu_instruction": round(n_salu / pmc["valu_insts_per_launch"], 3)}
        roofline = {"bound": "valu", "achieved": round(valu_achieved, 2), "peak": round(peak, 1),
                    "unit": "G wave64 VALU instr/s", "frac": round(valu_achieved / peak, 4), "peak_basis": basis,
                    "useful_frac": useful["useful_frac"] if useful else None, "useful_work": useful, "scalar_unit": scalar,
                    "traffic": pmc["traffic_bytes_per_launch"],
                    "valu_busy_fraction_pmc": pmc["valu_busy_fraction"], "avg_waves_per_simd": pmc["avg_waves_per_simd"],
                    # the metric counts nominal Ms*Mt tests; most are culled before any arithmetic:
                    "evaluated_pair_tests": {"prefilter": cand.get("prefilter_tests"), "exact": cand.get("exact_tests"),
                                             "prefilter_fraction_of_nominal": cand.get("prefilter_fraction_of_nominal")},
                    "hbm": hbm, **common}
    else:
        roofline = {"bound": "hbm", **{k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "traffic")},
                    "algorithmic_bytes": algo_bytes,
                    "note": "VALU-issue bound by design (< 0.2 B per pair test); the VALU roofline needs the PMC summary of "
                            "this build (tools/pmc_bench.sh)", **common}

    cpu_baseline, parity = None, {"config": args.config, "checked": False}
    if rank == 0 and world == 1:
        if args.parity_digest:
            parity = digest_parity(scene, args, l3d)
        if not args.no_cpu_baseline:
            cpu_baseline, live = run_cpu_baseline(scene, args, pair_tests, kNN, l3d)
            parity = parity if args.parity_digest else live
            if args.parity_digest:
                parity["live_sample"] = {k: live.get(k) for k in ("scene", "ok", "set_diff", "max_rel", "error")}

    # ---- another scene of the same size on a fresh context, after the first context has gone (a process that serves
    # one Line3D object per scene): what a second scene costs once the process is warm
    if cold is not None:
        l3d.close()
        # (three scenes, the median: one sample of a ~2 ms call on a box that has just run the CPU baseline on all its cores
        # varied between 1.7 and 4.3 ms on C1)
        samples = []
        for k in range(3):
            scene2 = make_config(args.config, seed=0x5EED0002 + k) if args.config != "C0" else scene
            ms2, tm2, g2 = cold_call(scene2, kNN, local_rank, stepper)
            samples.append((ms2, tm2))
            g2.close()
        samples.sort(key=lambda x: x[0])
        cold.update(second_scene_ms=round(samples[1][0], 3), second_scene_call=samples[1][1],
                    second_scene_samples_ms=[round(x[0], 3) for x in samples])

    # N > 1: BASELINE's 8-GPU configurations are C2 / C3 / C4 at fixed size -- the default line above is N rings of C1 (weak);
    # the C2 strong-scaling line is measured in the same process group and printed inside the one JSON line, so that the
    # first real multi-GPU run yields both (VERDICT r5 #6c).  Every rank takes part.
    extra_line = None
    if world > 1 and not args.no_extra_lines and args.config == "C1" and args.scaling == "weak":
        extra_line, _ = quick_line("C2", "strong", max(3, args.steps // 4), 1)
    if rank == 0:
        out = {
            "metric": "M segment-pair scores/sec", "value": round(value, 2), "unit": "M segment-pair scores/s",
            "n_gpus": world, "ranks_reported_by_backend": dist.get_world_size() if world > 1 else 1,
            "backend": dist.get_backend() if world > 1 else "none",
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if args.config != "C0" else "reference testdata (recovered cameras)",
            "config": {"workload": ((f"{world} rings of " if weak else "") +
                                    f"{args.config}: synthetic {cfg['n_views']} views x {cfg['n_segs']} segments/view, "
                                    if args.config != "C0" else
                                    "C0: the reference's bundled testdata (26 images, cameras recovered from its result "
                                    "fixture, the 406-919 LSD segments per view that occur in it), ") +
                                   f"{cfg['n_neighbors']} visual neighbours, kNN=10, sigma_p=2.5px, sigma_a=10deg, "
                                   f"epipolar_overlap=0.25; matchImages + affinity fill",
                       "pair_tests_per_step": pair_tests, "directed_pairs": len(pairs),
                       "views": scene.n_views,
                       "parallelism": (f"views / pairs sharded x{world} (halo form, tail and affinity fill sharded by views; "
                                       f"{'weak: one ring of the configuration per rank' if weak else 'strong: the same scene at every N'})")
                                      if world > 1 else "single GPU"},
            "phase_ms": {k: round(v / phase_steps, 4) for k, v in phase.items()},
            "cold_ms": cold["cold_ms"] if cold else None, "second_scene_ms": cold.get("second_scene_ms") if cold else None,
            # the call a user of Line3D::matchImages makes ONCE per scene, in a warm process: the second headline (round 6)
            "value_second_scene": round(pair_tests / (cold["second_scene_ms"] * 1e-3) / 1e6, 2) if cold and cold.get("second_scene_ms") else None,
            "second_scene_over_steady": round(cold["second_scene_ms"] / ms_per_step, 3) if cold and cold.get("second_scene_ms") else None,
            "cold": cold,
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
        }
        if extra_line is not None:
            out["baseline_8gpu_config_line"] = extra_line
        # (not in the profiling / A-B invocations, which all pass --no-cpu-baseline: a second configuration in the process would
        # mix its kernels into their per-kernel summaries)
        if world == 1 and args.config == "C1" and not args.no_extra_lines and not args.no_cpu_baseline:
            # BASELINE's 8-GPU configuration (C2, strong) on this one GPU, for the model of its N-GPU time (VERDICT r5 #6d)
            line2, (prs2, M2, ph2, ex2) = quick_line("C2", "strong", 3, 1)
            line2["multi_gpu_model"] = multi_gpu_model(prs2, M2, kNN, ph2, ph2["lists"], 1024.0 * ex2["record_kbytes"],
                                                       tail_bytes=48.0 * ex2["n_surv"] + 136.0 * ex2["n_best"] + 12.0 * sum(M2.values()),
                                                       sim_bytes=4.0 * ex2["n_surv"])
            out["baseline_8gpu_config_on_one_gpu"] = line2
        if world == 1:
            out["roofline_phase_b"] = phase_b_roofline(tm_last, lists_ms_sum / phase_steps, pairs, M, kNN, build, args.config, world)
        if world == 1:
            out["phase_ms"]["lists_part_of_finish"] = round(lists_ms_sum / phase_steps, 4)
            out["phase_ms"]["measured_in"] = (f"{phase_steps} separate untimed steps with all ten HIP events of a call on "
                                              "(l3d_set_timing_level 2); the timed steps record only the pair around the match kernel")
            # what the sharded tail / affinity fill exchange: the surviving matches (Match 40 B + two segment ids), the best
            # hypotheses (HypRec 128 B + a depth pair), three per-segment words; one float per surviving match
            out["multi_gpu_model"] = multi_gpu_model(pairs, M, kNN, {k: v / phase_steps for k, v in phase.items()},
                                                     lists_ms_sum / phase_steps, 1024.0 * record_kbytes,
                                                     tail_bytes=48.0 * n_surv + 136.0 * n_best + 12.0 * sum(M.values()),
                                                     sim_bytes=4.0 * n_surv)
        if world > 1 and getattr(l3d, "dist_ms", None):
            # rank 0's host wall time between the synchronisation points of the sharded call, per call (all calls incl.
            # warm-up): this rank's pairs | index all-gather | expansion + this rank's share of the list pass | all-gather
            # of its records | replicated remainder of phase B -- what the N-GPU time is made of
            calls = max(l3d.dist_ms.get("calls", 1), 1)
            out["phase_wall_ms"] = {k: round(v / calls, 4) for k, v in l3d.dist_ms.items() if k != "calls"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
