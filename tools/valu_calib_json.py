"""plain.json + PMC CSVs of tools/valu_calib.sh -> gpurun_out/<tag>_valu_calibration.json
   python tools/valu_calib_json.py <dir> <tag>"""
import csv
import glob
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
plain = json.load(open(os.path.join(d, "plain.json")))
pmc = {}
for f in glob.glob(d + "/p*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_"):
            continue
        pmc.setdefault(k[2:], {})[r["Counter_Name"]] = float(r["Counter_Value"])
ops = {}
for e in plain["ops"]:
    o = ops.setdefault(e["op"], {"op": e["op"], "class": e["class"], "unit": e["unit"], "by_waves_per_simd": {}})
    o["by_waves_per_simd"][e["waves_per_simd"]] = {k: e[k] for k in ("cycles_per_unit_simd", "cycles_per_unit_simd_wall", "clock_ghz",
                                                                     "G_units_per_s", "kernel_ms", "max_over_mean_wave_cycles")}
for name, o in ops.items():
    # The figure bench.py prices an instruction class with is the WALL-based one: all instructions of the launch / (kernel
    # time x measured shader clock x 1024 SIMDs), smallest over the wave counts.  The per-wave s_memtime figure
    # (cycles_per_unit_simd) divides a wave's own duration by the waves ASSUMED to share its SIMD: the dispatcher does not
    # spread 8192 one-wave workgroups evenly (max / mean wave time 1.3-1.9 at 8 waves per SIMD), so it reads low.
    o["cycles_per_unit_simd_best"] = min(v["cycles_per_unit_simd_wall"] for v in o["by_waves_per_simd"].values())
    o["cycles_per_unit_simd_per_wave_min"] = min(v["cycles_per_unit_simd"] for v in o["by_waves_per_simd"].values())
    c = pmc.get(name)
    if c:
        o["pmc_4_waves_per_simd"] = c
        nv = c.get("SQ_INSTS_VALU")
        if nv:
            # SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES count quad-cycles (MI355X_MICROARCH.md)
            o["SQ_ACTIVE_INST_VALU_quadcycles_per_SQ_INSTS_VALU"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / nv, 4)
            o["classes_counted_per_SQ_INSTS_VALU"] = {k[14:]: round(v / nv, 4) for k, v in c.items()
                                                      if k.startswith("SQ_INSTS_VALU_") and v}
out = {"_comment": "tools/valu_calib.hip on this box: cycles per wave64 instruction and SIMD for streams of one VALU instruction "
                   "(16 independent accumulators), from s_memtime per wave (cycles_per_unit_simd), from wall time x measured clock "
                   "(…_wall; cycles_per_unit_simd_best = its minimum over the wave counts, the figure that prices the roof), and the PMC counters of the same kernels at 4 waves per SIMD.  Streams that write VCC carry one "
                   "compiler-inserted s_nop per instruction.",
       "iters": plain["iters"], "ops": list(ops.values())}
path = os.path.join(d, "..", f"{tag}_valu_calibration.json")
json.dump(out, open(path, "w"), indent=1)
for o in out["ops"]:
    print(f'{o["op"]:16s} {o["class"][:28]:28s} best {o["cycles_per_unit_simd_best"]:8.3f} cyc/{o["unit"].split()[0]:9s}',
          {w: v["cycles_per_unit_simd_wall"] for w, v in o["by_waves_per_simd"].items()},
          o.get("SQ_ACTIVE_INST_VALU_quadcycles_per_SQ_INSTS_VALU"), o.get("classes_counted_per_SQ_INSTS_VALU"))
