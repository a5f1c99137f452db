"""rocprofv3 PMC CSVs of tools/pmc_bench.sh -> profiles/<tag>_pmc_match.json (per launch of k_match_pairs), keyed by
the library's build id:  python tools/pmc_to_json.py <dir with pass dirs> <config> <tag>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d, cfg, tag = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in sorted(glob.glob(d + "/*/**/p_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
per = {k: {c: v / max(len(disp[(k, c)]), 1) for c, v in cs.items()} for k, cs in acc.items()}
# the variant that did the work (bench.py's cold call / parity sample may launch other instantiations on small scenes)
name = max((k for k in per if k.startswith("k_match_pairs")), key=lambda k: acc[k].get("GRBM_GUI_ACTIVE", 0.0))
m = per[name]
from line3dpp_amd import _lib
# (reading the build id needs the library only, not a GPU)
build = _lib.load().l3d_build_info().decode()
cycles = m["GRBM_GUI_ACTIVE"] / 8.0                        # summed over the 8 XCDs
simds = 256 * 4
out = {
    "_comment": "rocprofv3 PMC passes of `bench.py --config %s --steps 2 --warmup 1` (tools/pmc_bench.sh), per launch of %s. "
                "SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; "
                "FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE counts half of a 16 B/lane stream on gfx950 (read bytes = 2 x "
                "FETCH_SIZE x 1024, MI355X_MICROARCH.md); WRITE_SIZE is uncalibrated there -- see profiles/r02_write_size_calibration.json "
                "(the calibration of round 2: the one that exists)" % (cfg, name),
    "build_info": build, "config": cfg, "kernel": name, "counters_per_launch": {k: round(v) for k, v in sorted(m.items())},
    "kernel_cycles": round(cycles),
    "valu_busy_fraction": round(4.0 * m["SQ_ACTIVE_INST_VALU"] / (cycles * simds), 4),
    "avg_waves_per_simd": round(4.0 * m["SQ_WAVE_CYCLES"] / (cycles * simds), 3),
    "valu_insts_per_launch": round(m["SQ_INSTS_VALU"]),
    # dynamic instruction mix (wave-instructions per launch by class; "other" = compares, selects, moves, min/max, lane ops)
    "valu_class_insts_per_launch": {k[14:]: round(v) for k, v in sorted(m.items()) if k.startswith("SQ_INSTS_VALU_")},
    "read_bytes_per_launch": round(2 * 1024 * m["FETCH_SIZE"]), "write_bytes_per_launch": round(1024 * m["WRITE_SIZE"]),
    "traffic_bytes_per_launch": round(2 * 1024 * m["FETCH_SIZE"] + 1024 * m["WRITE_SIZE"]),
    "other_kernels": {k: {c: round(v) for c, v in sorted(cs.items())} for k, cs in per.items() if k != name},
}
try:
    st = json.load(open(os.path.join(d, "stats.json")))
    if st.get("build_info") == build:
        out["candidates"] = {k: st.get(k) for k in ("nominal_pair_tests", "prefilter_tests", "exact_tests", "passed_overlap", "accepted",
                                                     "drains", "band_pairs", "kept_slots", "work_items",
                                                     "prefilter_fraction_of_nominal", "band_fraction_of_nominal")}
except Exception as e:   # noqa
    out["candidates"] = None
path = os.path.join(ROOT, "gpurun_out", "%s_pmc_match.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("build_info", "kernel", "valu_busy_fraction", "avg_waves_per_simd", "traffic_bytes_per_launch")}))
print("wrote", path, "-- copy it to profiles/")
