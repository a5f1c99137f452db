"""rocprofv3 PMC CSVs + kernel trace of tools/pmc_phase_b.sh -> gpurun_out/<tag>_pmc_lists_<config>.json: per kernel of one
`matchImages + affinity` step, per launch: duration (undisturbed trace), counted HBM-side traffic, instruction / wait counters.
   python tools/pmc_phase_b_json.py <dir with pass dirs> <config> <tag>"""
import collections
import csv
import glob
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d, cfg, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def short(name):
    """l3d::k_lists<1, 128>(args...) -> k_lists<1,128>"""
    n = name.split("(")[0]
    n = re.sub(r"^void\s+", "", n)
    n = n.replace("l3d::", "").replace("(anonymous namespace)::", "").replace(" ", "")
    return n


acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in sorted(glob.glob(d + "/[0-9]*/**/p_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
per = {k: {c: v / max(len(disp[(k, c)]), 1) for c, v in cs.items()} for k, cs in acc.items()}
launches = {k: max(len(disp[(k, c)]) for c in cs) for k, cs in acc.items()}

# undisturbed durations: the --kernel-trace --stats run (7 steps of bench.py + its 5 phase steps)
dur = {}
dbs = glob.glob(d + "/trace/**/*.db", recursive=True)
steps_in_trace = None
if dbs:
    db = sqlite3.connect(dbs[0])
    for name, n, tot, avg, mn in db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by name"):
        dur[short(name)] = {"calls": n, "avg_us": avg / 1e3, "min_us": mn / 1e3, "total_ms": tot / 1e6}
    m = [v["calls"] for k, v in dur.items() if k.startswith("k_match_pairs")]
    steps_in_trace = max(m) if m else None

from line3dpp_amd import _lib
build = _lib.load().l3d_build_info().decode()
kern = {}
for k, m in per.items():
    if not k.startswith("k_") or k.startswith("k_warm"):
        continue
    e = {"launches_profiled": launches[k]}
    if k in dur:
        e["avg_us"] = round(dur[k]["avg_us"], 2); e["min_us"] = round(dur[k]["min_us"], 2)
        e["calls_per_step"] = round(dur[k]["calls"] / steps_in_trace, 2) if steps_in_trace else None
    if "FETCH_SIZE" in m:
        # KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B (MI355X_MICROARCH.md): exact for wide coalesced
        # streams, an UPPER bound for 32/64-byte gathers -- both readings are given
        e["fetch_bytes_raw"] = round(1024 * m["FETCH_SIZE"]); e["fetch_bytes_x2"] = round(2 * 1024 * m["FETCH_SIZE"])
    if "WRITE_SIZE" in m:
        e["write_bytes"] = round(1024 * m["WRITE_SIZE"])
    if "TCC_HIT_sum" in m:
        e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1), 4)
    if "SQ_WAVES" in m:
        e["waves"] = round(m["SQ_WAVES"]); e["valu_insts"] = round(m["SQ_INSTS_VALU"]); e["salu_insts"] = round(m["SQ_INSTS_SALU"])
        e["lds_insts"] = round(m["SQ_INSTS_LDS"])
        e["wait_share_of_wave_life"] = round(m["SQ_WAIT_ANY"] / max(m["SQ_WAVE_CYCLES"], 1), 3)
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
        if cyc:
            e["kernel_cycles"] = round(cyc)
            e["valu_busy_fraction"] = round(4.0 * m["SQ_ACTIVE_INST_VALU"] / (cyc * 1024), 4)
            e["avg_waves_per_simd"] = round(4.0 * m["SQ_WAVE_CYCLES"] / (cyc * 1024), 3)
    if "SQ_INSTS_VMEM_RD" in m:
        e["vmem_rd_insts"] = round(m["SQ_INSTS_VMEM_RD"]); e["vmem_wr_insts"] = round(m["SQ_INSTS_VMEM_WR"]); e["smem_insts"] = round(m.get("SQ_INSTS_SMEM", 0))
        if m.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_share"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3)
        if "SQ_ACTIVE_INST_VALU" in m and m["SQ_ACTIVE_INST_VALU"]:
            e["lanes_of_16"] = round(m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"] / 4, 2)
    if "avg_us" in e and "fetch_bytes_x2" in e and "write_bytes" in e:
        t = e["avg_us"] * 1e-6
        e["counted_GB_per_s_x2"] = round((e["fetch_bytes_x2"] + e["write_bytes"]) / t / 1e9, 1)
        e["counted_GB_per_s_raw"] = round((e["fetch_bytes_raw"] + e["write_bytes"]) / t / 1e9, 1)
    kern[k] = e
out = {"_comment": "rocprofv3 PMC passes of `bench.py --config %s --steps 2 --warmup 1 --no-cpu-baseline --no-cold` (tools/pmc_phase_b.sh): per "
                   "LAUNCH of every kernel of a step; durations from a separate --kernel-trace run without counters.  FETCH_SIZE / WRITE_SIZE "
                   "in KiB; FETCH_SIZE x 2 is the guide's gfx950 correction for wide coalesced streams (an upper bound for narrow "
                   "gathers); WRITE_SIZE is uncalibrated beyond profiles/r02_write_size_calibration.json" % cfg,
       "build_info": build, "config": cfg, "kernels": dict(sorted(kern.items(), key=lambda kv: -(kv[1].get("avg_us", 0) * (kv[1].get("calls_per_step") or 1))))}
path = os.path.join(ROOT, "gpurun_out", "%s_pmc_lists_%s.json" % (tag, cfg))
json.dump(out, open(path, "w"), indent=1)
for k, e in list(out["kernels"].items())[:12]:
    print(f"{k[:40]:40s} {e.get('avg_us', 0):9.1f} us x{e.get('calls_per_step')}  fetch(x2) {e.get('fetch_bytes_x2', 0) / 1e6:9.1f} MB  write {e.get('write_bytes', 0) / 1e6:8.1f} MB  "
          f"{e.get('counted_GB_per_s_x2', 0):7.0f} GB/s  wait {e.get('wait_share_of_wave_life')}  busy {e.get('valu_busy_fraction')}  L2hit {e.get('l2_hit_rate')}")
print("wrote", path)
