"""Re-creates a runnable version of BASELINE config C0 (the reference's bundled testdata, 26 images).

The reference ships the images and its own RESULT (testdata/Line3D++_ref/...vis_3.txt: 2489 3D lines, each with its
residual 2D segments = (camID, segID, x1, y1, x2, y2)) but not the SfM input vsfm_result.nvm, so the cameras are
missing.  They are recovered here from the result itself: every residual 2D segment must contain the projection of
its 3D line, i.e. l^T P X = 0 for both end points X of the 3D line and the image line l through the 2D segment --
two linear equations per residual, several hundred per image -> P by DLT (Hartley normalisation, SVD), then
K [R|t] by RQ decomposition.  Median reprojection distance of the 3D end points to their 2D lines: 0.1-0.4 px.

Output tests/golden/real_scene_c0.npz: 26 views (K, R, t, 3072 x 2304), the 2D segments that occur in the fixture
(406-919 per view, renumbered compactly in ascending original segID), visual neighbours = the 10 views sharing
most 3D lines, per-view median depth, and the fixture's lines (residual sets in the compact numbering + end points)
for a plausibility check of reconstructions.  Run where /root/reference exists:
    python tests/golden/make_real_scene.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from line3dpp_amd.io import read_3d_lines_txt  # noqa: E402

REF_TXT = ("/root/reference/testdata/Line3D++_ref/"
           "Line3D++__W_FULL__N_10__sigmaP_2.5__sigmaA_10__epiOverlap_0.25__kNN_10__vis_3.txt")
WIDTH, HEIGHT = 3072, 2304      # testdata/img0000NN.jpg


def rq(M):
    Q, R = np.linalg.qr(np.flipud(M).T)
    R = np.fliplr(np.flipud(R.T)); Q = np.flipud(Q.T)
    S = np.diag(np.sign(np.diag(R)))
    return R @ S, S @ Q


def resect(obs):
    """obs: list of (coords4, P1, P2) -> K, R, t, median / 90th percentile point-to-line reprojection distance"""
    X = np.array([o[1] for o in obs] + [o[2] for o in obs]); mu = X.mean(0); sc = np.abs(X - mu).max()
    co = np.array([o[0] for o in obs]).reshape(-1, 2); pm = co.mean(0); ps = np.abs(co - pm).max()
    T3 = np.eye(4); T3[:3, :3] /= sc; T3[:3, 3] = -mu / sc
    T2 = np.array([[1 / ps, 0, -pm[0] / ps], [0, 1 / ps, -pm[1] / ps], [0, 0, 1]])
    A = []
    for c4, P1, P2 in obs:
        a = T2 @ np.array([c4[0], c4[1], 1.0]); b = T2 @ np.array([c4[2], c4[3], 1.0])
        ln = np.cross(a, b); ln /= np.linalg.norm(ln[:2])
        for P in (P1, P2):
            A.append(np.kron(ln, T3 @ np.append(P, 1.0)))
    P = np.linalg.svd(np.array(A))[2][-1].reshape(3, 4)
    P = np.linalg.inv(T2) @ P @ T3
    K, R = rq(P[:, :3])
    if np.linalg.det(R) < 0:
        P = -P
        K, R = rq(P[:, :3])
    t = np.linalg.inv(K) @ P[:, 3]
    K = K / K[2, 2]
    err = []
    for c4, P1, P2 in obs:
        ln = np.cross([c4[0], c4[1], 1.0], [c4[2], c4[3], 1.0]); ln /= np.linalg.norm(ln[:2])
        for Pw in (P1, P2):
            x = K @ (R @ Pw + t); err.append(abs(ln @ (x / x[2])))
    assert (R @ X.T + t[:, None])[2].min() > 0, "points behind the camera"
    return K, R, t, float(np.median(err)), float(np.percentile(err, 90))


def main():
    lines = read_3d_lines_txt(REF_TXT)
    cams = sorted({int(c) for L in lines for c in L["residuals"][:, 0]})
    obs = {c: [] for c in cams}; segs = {c: {} for c in cams}
    shared = np.zeros((max(cams) + 1, max(cams) + 1), int)
    for L in lines:
        P1, P2 = L["segments"][0][:3], L["segments"][-1][3:]
        seen = set()
        for (c, s), co in zip(L["residuals"], L["coords2D"]):
            obs[int(c)].append((co.astype(np.float64), P1, P2)); segs[int(c)][int(s)] = co; seen.add(int(c))
        for a in seen:
            for b in seen:
                shared[a, b] += a != b
    allP = np.concatenate([L["segments"].reshape(-1, 3) for L in lines])
    out = dict(cam=np.array(cams, np.uint32), K=[], R=[], t=[], median_depth=[], seg_off=[0], segs=[], orig_seg=[],
               nb_off=[0], nb=[], reproj_med_px=[], reproj_p90_px=[])
    idmap = {}
    for c in cams:
        K, R, t, e50, e90 = resect(obs[c])
        ids = sorted(segs[c]); idmap[c] = {s: i for i, s in enumerate(ids)}
        out["K"].append(K); out["R"].append(R); out["t"].append(t)
        out["median_depth"].append(np.median((R @ allP.T + t[:, None])[2]))
        out["segs"].append(np.array([segs[c][i] for i in ids], np.float32)); out["orig_seg"].append(np.array(ids, np.uint32))
        out["seg_off"].append(out["seg_off"][-1] + len(ids))
        nb = sorted(int(x) for x in np.argsort(-shared[c], kind="stable")[:10] if shared[c, x] > 0)
        out["nb"] += nb; out["nb_off"].append(len(out["nb"]))
        out["reproj_med_px"].append(e50); out["reproj_p90_px"].append(e90)
        print(f"cam {c:2d}: f = {K[0, 0]:.1f}/{K[1, 1]:.1f} pp = ({K[0, 2]:.1f}, {K[1, 2]:.1f}) segments {len(ids)} "
              f"reprojection {e50:.2f} px (p90 {e90:.2f})")
    fix_off, fix_res, fix_seg = [0], [], []
    for L in lines:
        fix_res += [(int(c), idmap[int(c)][int(s)]) for c, s in L["residuals"]]
        fix_off.append(len(fix_res)); fix_seg.append(np.concatenate([L["segments"][0][:3], L["segments"][-1][3:]]))
    res = dict(cam=out["cam"], K=np.array(out["K"]), R=np.array(out["R"]), t=np.array(out["t"]),
               median_depth=np.array(out["median_depth"], np.float32), seg_off=np.array(out["seg_off"], np.uint32),
               segs=np.concatenate(out["segs"]), orig_seg=np.concatenate(out["orig_seg"]),
               nb_off=np.array(out["nb_off"], np.uint32), nb=np.array(out["nb"], np.uint32),
               width=np.uint32(WIDTH), height=np.uint32(HEIGHT),
               reproj_med_px=np.array(out["reproj_med_px"], np.float32), reproj_p90_px=np.array(out["reproj_p90_px"], np.float32),
               fixture_res_off=np.array(fix_off, np.uint32), fixture_res=np.array(fix_res, np.uint32),
               fixture_endpoints=np.array(fix_seg, np.float32),
               source=np.array("cameras recovered by line-based DLT resection from testdata/Line3D++_ref/...kNN_10__vis_3.txt"))
    path = os.path.join(ROOT, "tests", "golden", "real_scene_c0.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
