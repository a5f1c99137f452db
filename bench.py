"""bench.py -- headline benchmark of the Line3D++ matching/scoring hot path on MI355X.

A "step" = one full pass of the hot path over one synthetic scene:
    Line3D::matchImages (pair matching -> orientation filter -> 3D scoring -> inverse matches ->
    hypothesis collapse) + the affinity fill of Line3D::reconstruct3Dlines.
Metric (BASELINE.json): million segment-pair scores per second = sum over matched directed view pairs
of Ms*Mt, divided by the wall time of the step, whole job over all ranks.  Segment arrays are resident
in HBM before the timed region (uploaded at addImage).  Workload at N=1: BASELINE config C1
(synthetic 64 views x 2000 segments/view, 10 visual neighbours).

N>1 (torchrun, one rank per GPU): the directed view pairs are sharded over the ranks, slot slices are
all-gathered over RCCL, the per-view chain is replicated ("scaling": "strong": same scene at every N).

Prints ONE JSON line with `roofline` (pair-matching kernel, HIP events on its launch stream) and
`cpu_baseline` (the CPU oracle = port of the reference OpenMP path, timed on this box, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def run_cpu_baseline(scene, args, pair_tests, kNN, l3d):
    """The reference's own OpenMP CPU path (oracle/_ref; the restatement where _ref is absent) on this box's host
    cores, and -- because that run IS the reference result for the benchmarked scene -- the parity check of the HIP
    result against it.  The reference's structure (std::list / std::map / priority_queue per row) stops scaling long
    before 256 threads, so the thread count is picked by a short scan on a sub-scene.  The sample is the full workload
    while it stays below ~6e9 pair tests (C0, C1, C3: 5-25 s); the larger configurations are sampled by a slice of
    consecutive views at the configured size and neighbour count.  Returns (cpu_baseline, parity)."""
    from oracle import oracle as O
    from tests import helpers as H
    # the reference's own line3D.cc/view.cc (oracle/_ref) in its Release configuration (-O3 -DNDEBUG, CMakeLists.txt:3;
    # tests/test_reference_pin.py: results byte-identical to the -O2 build the parity tests use), else the restatement
    use_ref = "release" if O.have_release() else O.have_reference()

    def Oracle(threads):
        return O.Oracle(threads=threads, reference=use_ref)
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
            o = Oracle(threads=t); o.add_scene(sub)
            t1 = time.perf_counter(); o.match_images(kNN=kNN); o.compute_affinity()
            scan[t] = round(sub_tests / (time.perf_counter() - t1) / 1e6, 1)
            if scan[t] < 0.5 * max(scan.values()):
                break                                    # past the knee: more threads only get slower
        best_t = max(scan, key=scan.get)
    sample, sample_tests, what = scene, pair_tests, f"full {args.config} workload once"
    if pair_tests > 6e9:
        n = scene.n_views
        while n > 4 and H.ring_slice(scene, 0, n).pair_tests()[0] > 5e9:
            n -= 1
        sample = H.ring_slice(scene, 0, n)
        sample_tests = sample.pair_tests()[0]
        what = f"slice of the first {n} views of {args.config} (configured segments/view and neighbour count) once"
    o = Oracle(threads=best_t); o.add_scene(sample)
    t1 = time.perf_counter()
    o.match_images(kNN=kNN); o.compute_affinity()
    cdt = time.perf_counter() - t1
    cpu = {"value": round(sample_tests / cdt / 1e6, 2), "unit": "M segment-pair scores/s", "cores": best_t,
           "kind": "reference" if use_ref else "port", "host_cpus": ncpu, "thread_scan_M_per_s": scan,
           "sample": f"{what} ({sample_tests} pair tests, {cdt:.2f} s) with the thread count that scored best on an "
                     f"8-view sub-scene; " +
                     ("the reference's own OpenMP CPU path (line3D.cc/view.cc compiled in place, oracle/_ref" +
                      (", -O3 -DNDEBUG)" if use_ref == "release" else ", -O2)") if use_ref
                      else "OpenMP oracle = restatement of the reference CPU path")}
    # ---- parity of the HIP result with that reference run (same scene, same parameters) ----
    if sample is scene:
        g = l3d                                          # the context of the timed steps holds the last step's result
    else:
        from line3dpp_amd.api import Line3D
        g = Line3D(device=torch.cuda.current_device()); g.add_scene(sample)
        assert g.matchImages(kNN=kNN) and g.computeAffinity()
    d = H.full_result_diff(g, o, sample)
    parity = {"config": args.config, "checked": True, "against": cpu["kind"], "scene": what, "ok": d["ok"],
              "surviving_matches": d["surviving"], "set_diff": d["set_diff"], "best_hypotheses": d["best"],
              "best_set_diff": d["best_set_diff"], "best_choice_diff": d.get("best_choice_diff"),
              "affinity_entries": d["affinity_entries"], "affinity_set_diff": d["affinity_set_diff"],
              "inexact_overlap_or_depth_fields": d["inexact_phase_a_fields"],
              "max_rel": max(d["max_rel_score3D"], d.get("max_rel_endpoints") or 0.0, d["max_rel_affinity"] or 0.0),
              "max_rel_score3D": d["max_rel_score3D"], "max_rel_endpoints": d.get("max_rel_endpoints"),
              "max_rel_affinity": d["max_rel_affinity"], "order_rows": d["order_rows"], "tie_rows": d["tie_rows"],
              "tolerance": H.REL_TOL}
    return cpu, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C1", help="BASELINE config: C0 | C1 | C2 | C3 | C4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all host cores)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from line3dpp_amd import dist as l3d_dist
    from line3dpp_amd.api import Line3D
    from line3dpp_amd.scene import CONFIGS, make_config

    scene = make_config(args.config)
    cfg = CONFIGS[args.config]
    l3d = Line3D(device=local_rank)
    l3d.add_scene(scene)            # segment arrays now resident in HBM
    pair_tests, pairs = scene.pair_tests()
    kNN = 10

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def step():
        ok = l3d_dist.match_images_sharded(l3d, rank, world, device=device, kNN=kNN)
        assert ok, "matchImages failed"
        assert l3d.computeAffinity(), "affinity failed"

    for _ in range(args.warmup):
        step()
    kern_ms, kern_launches, phase = 0.0, 0, dict(begin=0.0, match=0.0, finish=0.0, affinity=0.0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = l3d.timings()
        kern_ms += tm["match_kernel_ms"]; kern_launches += tm["match_kernel_launches"]
        phase["begin"] += tm["begin_ms"]; phase["match"] += tm["match_pairs_ms"]
        phase["finish"] += tm["finish_ms"]; phase["affinity"] += tm["affinity_ms"]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    value = pair_tests / (dt / args.steps) / 1e6

    # ---- roofline of the dominant kernel (k_match_pairs), this rank's launches ----
    M = {v.cam: len(v.segs) for v in scene.views}
    ranges = l3d_dist.pair_ranges([M[s] * M[t] for s, t in pairs], world)
    f, c = ranges[rank]
    my_pairs = pairs[f:f + c]
    # algorithmic bytes per launch (SURVEY.md §8d / DESIGN.md): per directed pair 16*(Ms+Mt) B of segment
    # records read + (32 + 4)*kNN*Ms B written: the result slots and, since the orientation filter is fused into
    # the kernel, the 4-byte inverse-list position of every slot (SURVEY's bound is 40*kNN*Ms)
    algo_bytes = sum(16 * (M[s] + M[t]) + 36 * kNN * M[s] for s, t in my_pairs)
    my_tests = sum(M[s] * M[t] for s, t in my_pairs)
    avg_ms = kern_ms / max(kern_launches, 1)
    # two waves share a 64-row work item while the launch has few of them (k_match.hip: match_waves_per_group)
    wpg = 2 if sum((M[s] + 63) // 64 for s, _ in my_pairs) <= 16384 else 1
    hbm_achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # What binds this kernel is VALU issue, not HBM (< 0.2 B per pair test): the primary roofline is the VALU one.
    # Its inputs are PMC counters, which cannot be read from inside this process: tools/pmc_bench.sh collects them
    # (rocprofv3 --pmc, separate passes) into profiles/*_pmc_match.json TOGETHER WITH THE BUILD ID of the library they
    # were measured on, and they are only quoted here when that id is the id of the library being timed now (and the
    # workload is the same); otherwise the block says so and falls back to the HBM figure, which is always live.
    from line3dpp_amd import _lib
    build = _lib.load().l3d_build_info().decode()
    pmc, pmc_note = None, "no PMC summary under profiles/ for this build"
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_match.json")), reverse=True):
        try:
            tr = json.load(open(path))
        except Exception:
            continue
        if tr.get("build_info") == build and tr.get("config") == args.config and world == 1:
            pmc, pmc_note = tr, os.path.relpath(path, ROOT) + " (rocprofv3 --pmc, same build id)"
            break
        pmc_note = f"{os.path.relpath(path, ROOT)} was measured on another build / workload ({tr.get('build_info')}, " \
                   f"{tr.get('config')}): not quoted"
    valu_peak = 256 * 4 * 2.4e9 / 4 / 1e9          # wave64 VALU instructions per second, all SIMDs (4 cycles each): G/s
    hbm = {"achieved": round(hbm_achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(hbm_achieved / 8000.0, 6),
           "algorithmic_bytes": algo_bytes, "traffic": pmc["traffic_bytes_per_launch"] if pmc else None}
    common = {"kernel": f"k_match_pairs<0,false,true,{wpg}>", "kernel_ms": round(avg_ms, 4),
              "nominal_pair_tests_per_s": round(my_tests / (avg_ms * 1e-3), 1) if avg_ms > 0 else 0.0,
              "pmc_source": pmc_note, "build": build}
    if pmc:
        valu_achieved = pmc["valu_insts_per_launch"] / (avg_ms * 1e-3) / 1e9
        cand = pmc.get("candidates") or {}
        roofline = {"bound": "valu", "achieved": round(valu_achieved, 2), "peak": round(valu_peak, 1),
                    "unit": "G wave64 VALU instr/s", "frac": round(valu_achieved / valu_peak, 4),
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
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline, parity = run_cpu_baseline(scene, args, pair_tests, kNN, l3d)

    if rank == 0:
        out = {
            "metric": "M segment-pair scores/sec", "value": round(value, 2), "unit": "M segment-pair scores/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if args.config != "C0" else "reference testdata (recovered cameras)",
            "config": {"workload": (f"{args.config}: synthetic {cfg['n_views']} views x {cfg['n_segs']} segments/view, "
                                    if args.config != "C0" else
                                    "C0: the reference's bundled testdata (26 images, cameras recovered from its result "
                                    "fixture, the 406-919 LSD segments per view that occur in it), ") +
                                   f"{cfg['n_neighbors']} visual neighbours, kNN=10, sigma_p=2.5px, sigma_a=10deg, "
                                   f"epipolar_overlap=0.25; matchImages + affinity fill",
                       "pair_tests_per_step": pair_tests, "directed_pairs": len(pairs),
                       "parallelism": f"pair-sharded x{world}" if world > 1 else "single GPU"},
            "phase_ms": {k: round(v / args.steps, 4) for k, v in phase.items()},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
        }
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
