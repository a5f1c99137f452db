"""Per-work-item timeline of k_match_pairs from the diagnostic build (-DL3D_STATS, tools/build_stats_lib.sh):
   L3D_LIB=gpurun_scratch/libl3dpp_hip_stats.so python tools/cycles_run.py C1
Every work item's first wave records (shader cycles) its start, the duration of its main loop, the time inside the two
stages of the candidate pipeline, the epilogue and where it ran.  Printed: the distribution of the item durations, the
share of each part, how the launch fills the machine over time (items in flight per decile of the launch) and the tail.
NOTE: the counters of the diagnostic build (atomics in the hot loop) slow the kernel down ~2x: shares, not absolute times."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from line3dpp_amd import _lib  # noqa: E402
from line3dpp_amd.api import Line3D  # noqa: E402
from line3dpp_amd.scene import make_config  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
sc = make_config(cfg)
g = Line3D(); g.add_scene(sc)
L = _lib.load()
assert g.matchBegin() and g.matchPairs(0, len(g.pairs()[0]))      # warm
g.matchAbort()
assert g.matchBegin() and g.matchPairs(0, len(g.pairs()[0]))
tm = g.timings()
n_items = sum((len(v.segs) + 63) // 64 for v in sc.views for _ in range(1))  # per pair below
pairs = g.pairs()[0]
M = {v.cam: len(v.segs) for v in sc.views}
n_items = 1 << 16   # (the kernel records its first 65 536 work items; padding items of the width-class layout record nothing)
buf = np.zeros((n_items, 8), np.uint64)
L.l3d_debug_cycles(buf.ctypes.data_as(C.c_void_p), n_items)
start, loop, s1, s2, epi, total = (buf[:, k].astype(np.float64) for k in range(6))
wend = buf[:, 6].astype(np.float64)
ok = total > 0
start, loop, s1, s2, epi, total, wend = (a[ok] for a in (start, loop, s1, s2, epi, total, wend))
# start / wend: 100 MHz wall counter (common to the whole device); the rest in shader cycles of the item's own CU
t0, t1 = start.min(), wend.max()
span = t1 - t0
dur = wend - start
out = {"config": cfg, "build_info": L.l3d_build_info().decode(), "items_recorded": int(ok.sum()), "match_kernel_ms": tm["match_kernel_ms"],
       "launch_span_us": float(span) / 100.0,
       "item_us_pct_5_50_95_max": [float(x) / 100.0 for x in np.percentile(dur, [5, 50, 95, 100])],
       "item_cycles_pct_5_50_95_max": [float(x) for x in np.percentile(total, [5, 50, 95, 100])],
       "share_of_item_time": {"walk (pre-filter, pushes)": float(((loop - s1 - s2).sum()) / total.sum()), "stage 1 (depth decision)": float(s1.sum() / total.sum()),
                              "stage 2 (exact overlap + insertion)": float(s2.sum() / total.sum()), "epilogue": float(epi.sum() / total.sum())},
       "mean_items_in_flight": float(dur.sum() / span)}
total = dur
# items in flight over the launch, by decile
edges = np.linspace(t0, t1, 11)
infl = []
for a, b in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(start + total, b) - np.maximum(start, a), 0, None).sum() / (b - a)
    infl.append(round(float(ov), 1))
out["items_in_flight_by_decile_of_the_launch"] = infl
out["last_item_start_fraction_of_span"] = float((start.max() - t0) / span)
print(json.dumps(out))
