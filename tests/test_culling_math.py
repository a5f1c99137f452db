"""CPU check of the geometry behind the epipolar-band culling of phase A (k_match.hip / make_cull in
l3d_api.hip): a numpy restatement of the tau parametrisation and of the band rules, tested against the oracle --
every (source, target) segment pair the reference's matchingCPU accepts at a near-zero overlap threshold must have
intersecting bands.  (That the HIP kernels implement these rules is the business of the GPU tests, which compare the
culled path with the brute-force path and with the oracle; this file pins the claim the rules rest on.)"""
import numpy as np
import pytest

from line3dpp_amd.scene import make_scene
from oracle.oracle import Oracle


def _fundamental(s, t):                       # Line3D::getFundamentalMatrix, line3D.cc:874-892
    R = t.R @ s.R.T
    tt = t.t - R @ s.t
    T = np.array([[0, -tt[2], tt[1]], [tt[2], 0, -tt[0]], [-tt[1], tt[0], 0]])
    return np.linalg.inv(t.K.T) @ (T @ R) @ np.linalg.inv(s.K)


def _cull_forms(F, ws, hs, wt, ht):
    """As, Bs, At, Bt of make_cull, or None where the host refuses to cull the pair"""
    cols = [F[:, j] for j in range(3)]
    e, best = None, 0.0
    for a, b in ((0, 1), (0, 2), (1, 2)):
        x = np.cross(cols[a], cols[b]); n = np.linalg.norm(x)
        if n > best:
            best, e = n, x / n
    c = np.array([0.5 * wt, 0.5 * ht, 1.0])
    m = np.array([e[0] - c[0] * e[2], e[1] - c[1] * e[2]])
    if np.linalg.norm(m) < 1e-12:
        return None
    m /= np.linalg.norm(m)
    n = np.array([-m[1], m[0], 0.0])
    At, Bt = np.cross(e, c), np.cross(n, e)
    st = Bt @ c
    At, Bt = At / st, Bt / st
    As, Bs = -(F.T @ c), F.T @ n
    ss = Bs @ np.array([0.5 * ws, 0.5 * hs, 1.0])
    As, Bs = As / ss, Bs / ss

    def ok(B, w, h):                          # one sign, with margin, over the (slightly enlarged) image
        return all(B[0] * x + B[1] * y + B[2] >= 0.1 for x in (-0.05 * w, 1.05 * w) for y in (-0.05 * h, 1.05 * h))
    return (As, Bs, At, Bt) if ok(Bt, wt, ht) and ok(Bs, ws, hs) else None


def _bands(forms, S, T):
    As, Bs, At, Bt = forms

    def tau(A, B, x, y):
        return (A[0] * x + A[1] * y + A[2]) / (B[0] * x + B[1] * y + B[2])
    S = S.astype(np.float64); T = T.astype(np.float64)
    s1, s2 = tau(As, Bs, S[:, 0], S[:, 1]), tau(As, Bs, S[:, 2], S[:, 3])
    slo, shi = np.minimum(s1, s2), np.maximum(s1, s2)
    t1, t2 = tau(At, Bt, T[:, 0], T[:, 1]), tau(At, Bt, T[:, 2], T[:, 3])
    tlo, thi = np.minimum(t1, t2), np.maximum(t1, t2)
    # widening: the pencil line parallel to the target segment (tau of its direction), when it can lie in a wedge
    dx, dy = T[:, 2] - T[:, 0], T[:, 3] - T[:, 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        td = (At[0] * dx + At[1] * dy) / (Bt[0] * dx + Bt[1] * dy)
    inside = (td >= slo.min() - 1.0) & (td <= shi.max() + 1.0)
    tlo = np.where(inside, np.minimum(tlo, td), tlo); thi = np.where(inside, np.maximum(thi, td), thi)
    pad = lambda v: 0.01 + 1e-6 * np.abs(v)
    return slo - pad(slo), shi + pad(shi), tlo - pad(tlo), thi + pad(thi)


@pytest.mark.parametrize("n_views,n_segs,nn,seed,radius", [(12, 300, 4, 3, 25.0), (16, 400, 6, 11, 40.0),
                                                            (9, 500, 4, 29, 14.0)])
def test_accepted_matches_lie_in_intersecting_bands(n_views, n_segs, nn, seed, radius):
    sc = make_scene(n_views, n_segs, n_neighbors=nn, seed=seed, radius=radius)
    V = {v.cam: v for v in sc.views}
    o = Oracle(threads=4); o.add_scene(sc)
    o.begin_match(kNN=0, epi_overlap=1e-6)         # keep every match whose overlap exceeds (almost) zero
    _, pairs = sc.pair_tests()
    n_checked = n_culled_pairs = 0
    kept, total = 0, 0
    for s, t in pairs:
        forms = _cull_forms(_fundamental(V[s], V[t]), V[s].width, V[s].height, V[t].width, V[t].height)
        if forms is None:
            continue                               # the product streams such a pair unculled
        n_culled_pairs += 1
        slo, shi, tlo, thi = _bands(forms, V[s].segs, V[t].segs)
        m, _ = o.match_pair(s, t)
        r, q = m["src_seg"].astype(np.int64), m["tgt_seg"].astype(np.int64)
        inter = (tlo[q] <= shi[r]) & (thi[q] >= slo[r])
        assert inter.all(), (s, t, int((~inter).sum()), m[~inter][:3])
        n_checked += len(m)
        kept += int(((tlo[None, :] <= shi[:, None]) & (thi[None, :] >= slo[:, None])).sum()); total += len(slo) * len(tlo)
    o.end_match()
    assert n_culled_pairs > len(pairs) // 2 and n_checked > 1000
    assert kept / total < 0.5                      # the bands do cull: well under half of all pairs remain


def test_direction_widening_is_necessary():
    """Control: with the plain bands [tau(q1), tau(q2)] alone some accepted matches fall outside (a wedge that contains
    the target's own direction meets its line on both sides of the segment) -- the widening rule is not optional."""
    sc = make_scene(16, 400, n_neighbors=6, seed=11, radius=40.0)
    V = {v.cam: v for v in sc.views}
    o = Oracle(threads=4); o.add_scene(sc)
    o.begin_match(kNN=0, epi_overlap=1e-6)
    outside = total = 0
    for s, t in sc.pair_tests()[1]:
        forms = _cull_forms(_fundamental(V[s], V[t]), V[s].width, V[s].height, V[t].width, V[t].height)
        if forms is None:
            continue
        As, Bs, At, Bt = forms
        S, T = V[s].segs.astype(np.float64), V[t].segs.astype(np.float64)
        tau = lambda A, B, x, y: (A[0] * x + A[1] * y + A[2]) / (B[0] * x + B[1] * y + B[2])
        s1, s2 = tau(As, Bs, S[:, 0], S[:, 1]), tau(As, Bs, S[:, 2], S[:, 3])
        t1, t2 = tau(At, Bt, T[:, 0], T[:, 1]), tau(At, Bt, T[:, 2], T[:, 3])
        slo, shi = np.minimum(s1, s2) - 0.01, np.maximum(s1, s2) + 0.01
        tlo, thi = np.minimum(t1, t2) - 0.01, np.maximum(t1, t2) + 0.01
        m, _ = o.match_pair(s, t)
        r, q = m["src_seg"].astype(np.int64), m["tgt_seg"].astype(np.int64)
        outside += int((~((tlo[q] <= shi[r]) & (thi[q] >= slo[r]))).sum()); total += len(m)
    o.end_match()
    assert total > 100000 and 0 < outside < total // 100
