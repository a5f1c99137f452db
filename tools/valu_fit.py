"""Least-squares fit of the per-unit VALU instruction costs of k_match_pairs (tools/valu_fit_run.py):
   python tools/valu_fit.py <dir> <tag>  ->  gpurun_out/<tag>_valu_fit.json"""
import csv
import glob
import json
import os
import sys

import numpy as np

d, tag = sys.argv[1], sys.argv[2]
st = json.load(open(os.path.join(d, "stats.json")))
pr = json.load(open(os.path.join(d, "product.json")))
disp = {}
for f in glob.glob(d + "/p/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_match_pairs" in r["Kernel_Name"]:
            disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"].split("(")[0].split("::")[-1]})[r["Counter_Name"]] = float(r["Counter_Value"])
ds = [disp[k] for k in sorted(disp)]
runs = st["runs"]
assert len(ds) == len(runs), (len(ds), len(runs))
K = [r["kNN"] for r in runs]
# units: target visits of a wave (64 pre-filter tests each), drains (up to 64 exact tests each), epilogue passes of a wave
# (64 (row, slot) items each: 64 rows x kNN items per work item = kNN passes, whatever the waves per item), waves
# (the stage-1 drains of the two-stage pipeline are proportional to the stage-2 drains on one scene -- the share of
# candidates that pass the depth decision is a property of the scene -- so they are not a term of their own: the
# coefficient of a stage-2 drain carries the stage-1 work that feeds it)
A = np.array([[r["prefilter_tests"] / 64.0, r["drains"], r["work_items"] * k, d_["SQ_WAVES"]] for r, k, d_ in zip(runs, K, ds)])
y = np.array([d_["SQ_INSTS_VALU"] for d_ in ds])
x, res, rank, sv = np.linalg.lstsq(A, y, rcond=None)
pred = A @ x
out = {"_comment": "k_match_pairs: SQ_INSTS_VALU (product build, rocprofv3) of 12 runs of one 16-view x 2000-segment scene "
                   "(epipolar-overlap threshold x kNN) fitted as a*target_visits + b*drains (a stage-2 drain = exact overlap + insertion of 64 "
                   "candidates, with the stage-1 depth decisions that feed it) + e*epilogue_passes + w*waves; "
                   "unit counts from the -DL3D_STATS build of the same sources",
       "build_info": pr["build_info"], "stats_build_info": st["build_info"],
       "valu_per_target_visit": round(float(x[0]), 2),
       "stage1_drains_per_drain": round(float(np.mean([r.get("stage1_drains", 0) / max(r["drains"], 1) for r in runs])), 3),
       "valu_per_drain": round(float(x[-3]), 1),
       "valu_per_epilogue_pass": round(float(x[-2]), 1), "valu_per_wave_fixed": round(float(x[-1]), 1),
       "max_rel_residual": round(float(np.max(np.abs(pred - y) / y)), 4),
       "runs": [dict(r, **{"SQ_INSTS_VALU": d_["SQ_INSTS_VALU"], "SQ_WAVES": d_["SQ_WAVES"], "kernel": d_["kernel"],
                           "fit": round(float(p))}) for r, d_, p in zip(runs, ds, pred)]}
path = os.path.join(d, "..", f"{tag}_valu_fit.json")
json.dump(out, open(path, "w"), indent=1)
print({k: v for k, v in out.items() if k not in ("runs", "_comment")})
