"""Comparison helpers shared by the parity tests: HIP path (line3dpp_amd.Line3D) vs CPU oracle."""
import numpy as np

from line3dpp_amd._lib import EMPTY

REL_TOL = 1e-4   # north_star: results within 1e-4 relative of the reference CPU path


def rel_close(a, b, tol=REL_TOL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b) <= tol * np.maximum(np.abs(a), np.abs(b)) + 1e-30


def slots_to_dict(slots):
    """(src_seg, tgt_seg) -> (overlap, dp1, dp2, dq1, dq2) for the valid slots of one pair; also row order"""
    out = {}
    rows = {}
    Ms, K = slots.shape
    for r in range(Ms):
        row = slots[r]
        valid = row["tgt_seg"] != EMPTY
        # valid slots must be a prefix
        nv = int(valid.sum())
        assert valid[:nv].all(), f"row {r}: valid slots are not a prefix"
        rows[r] = [int(x) for x in row["tgt_seg"][:nv]]
        for s in row[:nv]:
            out[(r, int(s["tgt_seg"]))] = (s["overlap"], s["d_p1"], s["d_p2"], s["d_q1"], s["d_q2"])
    return out, rows


def oracle_pair_to_dict(matches):
    out = {}
    rows = {}
    for m in matches:
        out[(int(m["src_seg"]), int(m["tgt_seg"]))] = (m["overlap"], m["d_p1"], m["d_p2"], m["d_q1"], m["d_q2"])
        rows.setdefault(int(m["src_seg"]), []).append(int(m["tgt_seg"]))
    return out, rows


def compare_pair(slots, omatches):
    """returns dict(n_gpu, n_cpu, missing, extra, max_rel, bit_exact, order_mismatch)"""
    g, grows = slots_to_dict(slots)
    o, orows = oracle_pair_to_dict(omatches)
    missing = [k for k in o if k not in g]
    extra = [k for k in g if k not in o]
    max_rel = 0.0
    bit_exact = True
    for k in o:
        if k in g:
            a = np.array(g[k], np.float64); b = np.array(o[k], np.float64)
            if not np.array_equal(np.array(g[k], np.float32), np.array(o[k], np.float32)):
                bit_exact = False
            max_rel = max(max_rel, float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))))
    order_mismatch = sum(1 for r in orows if r in grows and grows[r] != orows[r])
    return dict(n_gpu=len(g), n_cpu=len(o), missing=missing, extra=extra, max_rel=max_rel, bit_exact=bit_exact,
                order_mismatch=order_mismatch)


def matches_to_dict(m):
    return {(int(x["src_seg"]), int(x["tgt_cam"]), int(x["tgt_seg"])): x for x in m}


def compare_matches(gm, om):
    g = matches_to_dict(gm); o = matches_to_dict(om)
    missing = [k for k in o if k not in g]
    extra = [k for k in g if k not in o]
    max_rel = 0.0
    for k in o:
        if k in g:
            for f in ("overlap", "score3D", "d_p1", "d_p2", "d_q1", "d_q2"):
                a, b = float(g[k][f]), float(o[k][f])
                max_rel = max(max_rel, abs(a - b) / max(abs(b), 1e-30))
    return dict(n_gpu=len(g), n_cpu=len(o), missing=missing, extra=extra, max_rel=max_rel)


def affinity_map(edges, l2g, key=lambda r: (int(r[0]), int(r[1]))):
    """{unordered (Segment2D, Segment2D) -> w}; also checks the (i,j),(j,i) pairing of A_"""
    out = {}
    assert len(edges) % 2 == 0
    for k in range(0, len(edges), 2):
        e1, e2 = edges[k], edges[k + 1]
        assert e1["i"] == e2["j"] and e1["j"] == e2["i"] and e1["w"] == e2["w"]
        a = key(l2g[e1["i"]]); b = key(l2g[e1["j"]])
        kk = (a, b) if a < b else (b, a)
        assert kk not in out, "duplicate unordered pair in A_"
        out[kk] = float(e1["w"])
    return out


def split_scene(sc, frac=0.5):
    """Breaks the first `frac` of every view's segments into two collinear halves with a gap (same segment count):
    the fragmented lines View::findCollinearSegments is meant for."""
    for v in sc.views:
        s = v.segs.astype(np.float64); n = int(len(s) * frac)
        a = s[:n, :2]; b = s[:n, 2:]
        h1 = np.concatenate([a, a + 0.45 * (b - a)], 1); h2 = np.concatenate([a + 0.55 * (b - a), b], 1)
        v.segs = np.concatenate([h1, h2, s[n:len(s) - n]], 0).astype(np.float32)
    return sc
