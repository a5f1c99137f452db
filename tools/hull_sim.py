"""CPU model of the walk of k_match_pairs: lane-tests of the fp32 pre-filter for row groups of R rows (numpy restatement
of the band rules, tests/test_culling_math.py), against the (row, target) pairs whose own bands intersect.
usage: python tools/hull_sim.py C1 [n_pairs]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from line3dpp_amd.scene import make_config
from tests.test_culling_math import _fundamental, _cull_forms, _bands

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sc = make_config(cfg)
V = {v.cam: v for v in sc.views}
_, pairs = sc.pair_tests()
rng = np.random.default_rng(1)
sel = rng.choice(len(pairs), size=min(npairs, len(pairs)), replace=False)
Rs = [64, 32, 16, 8]
tot = {R: 0 for R in Rs}; cv = {R: 0 for R in Rs}; useful = 0; nominal = 0
for pi in sel:
    s, t = pairs[pi]
    forms = _cull_forms(_fundamental(V[s], V[t]), V[s].width, V[s].height, V[t].width, V[t].height)
    if forms is None:
        continue
    S, T = V[s].segs, V[t].segs
    slo, shi, tlo, thi = _bands(forms, S, T)
    Ms, Mt = len(S), len(T)
    nominal += Ms * Mt
    # row order: width classes for large views (k_cull_prepare), then lo
    ref = 0.5 * (V[t].width + V[t].height)
    w = shi - slo
    two = Ms >= 8192
    w1 = ref * (1 / 32 if two else 1 / 16) if Ms >= 4096 else np.inf
    w2 = ref / 8
    cls = np.where(w > w1, np.where(two & (w > w2), 1, 2), 3)
    order = np.lexsort((slo, cls))
    slo, shi = slo[order], shi[order]
    # targets: (widened first), lo
    As, Bs, At, Bt = forms
    Tt = T.astype(np.float64)
    dx, dy = Tt[:, 2] - Tt[:, 0], Tt[:, 3] - Tt[:, 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        td = (At[0] * dx + At[1] * dy) / (Bt[0] * dx + Bt[1] * dy)
    tcls = np.where((td >= slo.min() - 1.0) & (td <= shi.max() + 1.0), 1, 2)
    to = np.lexsort((tlo, tcls))
    tlo, thi = tlo[to], thi[to]
    # useful: per row count of intersecting targets
    # sort-based counting: count targets with tlo <= shi_r  minus targets with thi < slo_r
    thi_sorted = np.sort(thi); tlo_sorted = np.sort(tlo)
    useful += int((np.searchsorted(tlo_sorted, shi, side="right") - np.searchsorted(thi_sorted, slo, side="left")).sum())
    nch = (Mt + 63) // 64
    pad = nch * 64 - Mt
    clo = np.pad(tlo, (0, pad), constant_values=np.inf).reshape(nch, 64).min(1)
    chi = np.pad(thi, (0, pad), constant_values=-np.inf).reshape(nch, 64).max(1)
    for R in Rs:
        for g0 in range(0, Ms, R):
            lo, hi = slo[g0:g0 + R].min(), shi[g0:g0 + R].max()
            n = int(np.count_nonzero(~((thi < lo) | (tlo > hi))))
            tot[R] += R * n
            cv[R] += int(np.count_nonzero(~((chi < lo) | (clo > hi))))
print(cfg, "pairs", len(sel), "nominal", nominal, "useful(band pairs)", useful, f"= {useful / nominal:.4f} of nominal")
for R in Rs:
    print(f"R={R:3d}: lane-tests {tot[R]:>14d}  = {tot[R] / nominal:.4f} of nominal, {tot[R] / useful:.2f} x useful; chunk visits {cv[R]} "
          f"({tot[R] / 64 / max(cv[R], 1):.1f} target-steps of 64 lanes per chunk visit)")
