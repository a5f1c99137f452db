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


# ---- vectorised whole-scene comparison (BASELINE-size scenes: millions of matches, no Python loops per match) ----
def _match_keys(m):
    return (m["src_seg"].astype(np.uint64) << np.uint64(40)) | (m["tgt_cam"].astype(np.uint64) << np.uint64(20)) | \
        m["tgt_seg"].astype(np.uint64)


def _max_rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30)))


def _aff_table(edges, l2g_keys):
    """sorted [(lo key, hi key)] array + weights of the unordered pairs of A_ (edges come as (i,j),(j,i) pairs)"""
    e = edges[0::2]
    a = l2g_keys[e["i"]]; b = l2g_keys[e["j"]]
    lo = np.minimum(a, b); hi = np.maximum(a, b)
    order = np.lexsort((hi, lo))
    return lo[order], hi[order], e["w"][order]


def full_result_diff(g, o, scene):
    """HIP context `g` (line3dpp_amd.Line3D after matchImages + computeAffinity) against oracle `o` (oracle.Oracle
    after match_images + compute_affinity), every view of `scene`, vectorised.  Surviving match lists are compared as
    sets AND in order (a list is a per-segment private container in the reference, so its order is deterministic even
    with OpenMP); best hypotheses and A_ are compared as maps (their global order is scheduling-dependent when the
    oracle runs multi-threaded).  Returns a dict of counts / maxima; `ok` summarises the north_star bar (identical
    sets, values within REL_TOL)."""
    r = dict(surviving=0, set_diff=0, order_rows=0, tie_rows=0, inexact_phase_a_fields=0, max_rel_score3D=0.0,
             views=len(scene.views))
    for v in scene.views:
        gm, goff = g.matches(v.cam); om, ooff = o.matches(v.cam)
        gk, ok_ = _match_keys(gm), _match_keys(om)
        r["surviving"] += len(om)
        same_order = len(gk) == len(ok_) and np.array_equal(gk, ok_)
        if not same_order:
            gs, os_ = np.sort(gk), np.sort(ok_)
            if len(gs) != len(os_) or not np.array_equal(gs, os_):
                r["set_diff"] += len(np.setxor1d(gk, ok_))
                common = np.intersect1d(gk, ok_)
                gm = gm[np.isin(gk, common)]; om = om[np.isin(ok_, common)]
                gk, ok_ = _match_keys(gm), _match_keys(om)
            else:
                # same set, different order inside some lists: which segments, and is every one of them an exact
                # overlap tie (the only order freedom: libstdc++ priority_queue pop order of equal keys)?
                segs = np.unique(gm["src_seg"][gk != ok_])
                r["order_rows"] += len(segs)
                for s in segs:
                    a = gm[goff[s]:goff[s + 1]]; b = om[ooff[s]:ooff[s + 1]]
                    d = np.nonzero(_match_keys(a) != _match_keys(b))[0]
                    if np.array_equal(a["overlap"][d], b["overlap"][d]):
                        r["tie_rows"] += 1
            gm = gm[np.argsort(gk, kind="stable")]; om = om[np.argsort(ok_, kind="stable")]
        for f in ("overlap", "d_p1", "d_p2", "d_q1", "d_q2"):
            r["inexact_phase_a_fields"] += int(np.count_nonzero(gm[f] != om[f]))
        r["max_rel_score3D"] = max(r["max_rel_score3D"], _max_rel(gm["score3D"], om["score3D"]))
        gi, oi = g.view_info(v.cam), o.view_info(v.cam)
        r["k_mismatch"] = r.get("k_mismatch", 0) + int(gi["k"] != oi["k"])
        r["max_rel_median_depth"] = max(r.get("max_rel_median_depth", 0.0), _max_rel(gi["median_depth"], oi["median_depth"]))
    # best hypotheses (estimated_position3D_) as a map (cam, seg) -> (P1, P2, dir, best match)
    s2, s3, bm = g.best(); cs, geo, _, obm = o.best()
    gkey = (s2["cam"].astype(np.uint64) << np.uint64(32)) | s2["seg"].astype(np.uint64)
    okey = (cs[:, 0].astype(np.uint64) << np.uint64(32)) | cs[:, 1].astype(np.uint64)
    r["best"] = len(okey)
    r["best_set_diff"] = len(np.setxor1d(gkey, okey))
    if r["best_set_diff"] == 0:
        gi_, oi_ = np.argsort(gkey), np.argsort(okey)
        gg = np.concatenate([s3["P1"], s3["P2"], s3["dir"]], 1)[gi_]
        og = geo[oi_]
        # per 3-vector: |difference| relative to the vector's length (a coordinate near zero has no relative scale)
        err = 0.0
        for c in (0, 3, 6):
            den = np.maximum(np.linalg.norm(og[:, c:c + 3], axis=1), 1e-30)
            err = max(err, float(np.max(np.linalg.norm(gg[:, c:c + 3] - og[:, c:c + 3], axis=1) / den))) if len(og) else err
        r["max_rel_endpoints"] = err
        r["best_choice_diff"] = int(np.count_nonzero((bm["tgt_cam"][gi_] != obm["tgt_cam"][oi_]) |
                                                     (bm["tgt_seg"][gi_] != obm["tgt_seg"][oi_])))
    # A_ as a map {unordered (Segment2D, Segment2D) -> w}
    ge, gl, gms = g.affinity(); oe, ol = o.affinity()
    glk = (gl["cam"].astype(np.uint64) << np.uint64(32)) | gl["seg"].astype(np.uint64)
    olk = (ol[:, 0].astype(np.uint64) << np.uint64(32)) | ol[:, 1].astype(np.uint64)
    ga, gb, gw = _aff_table(ge, glk); oa, ob, ow = _aff_table(oe, olk)
    r["affinity_entries"] = len(oe)
    same = len(ga) == len(oa) and np.array_equal(ga, oa) and np.array_equal(gb, ob)
    r["affinity_set_diff"] = 0 if same else int(len(set(zip(ga.tolist(), gb.tolist())) ^ set(zip(oa.tolist(), ob.tolist()))))
    r["max_rel_affinity"] = _max_rel(gw, ow) if same else None
    r["max_rel_med_scene_depth"] = _max_rel(gms, o.med_scene_depth_lines())
    r["ok"] = bool(r["set_diff"] == 0 and r["best_set_diff"] == 0 and r["affinity_set_diff"] == 0 and
                   r["inexact_phase_a_fields"] == 0 and r.get("best_choice_diff", 1) == 0 and r["k_mismatch"] == 0 and
                   r["order_rows"] == r["tie_rows"] and
                   max(r["max_rel_score3D"], r.get("max_rel_endpoints", 1.0), r["max_rel_affinity"] or 0.0,
                       r["max_rel_median_depth"], r["max_rel_med_scene_depth"]) <= REL_TOL)
    return r


def ring_slice(scene, first, count):
    """`count` consecutive views of a ring scene starting at view `first`, neighbours restricted to the slice: the
    views in the middle of the slice keep their full neighbour set, i.e. the configured neighbour count x size."""
    import copy
    keep = {scene.views[(first + i) % scene.n_views].cam for i in range(count)}
    out = copy.copy(scene)
    out.views = []
    for i in range(count):
        w = copy.copy(scene.views[(first + i) % scene.n_views])
        w.neighbors = [n for n in w.neighbors if n in keep] or \
            [scene.views[(first + (i + 1) % count) % scene.n_views].cam]   # never empty (line3D.cc:154 rejects the view)
        out.views.append(w)
    out.views.sort(key=lambda v: v.cam)
    out.name = f"{scene.name}[{first}:{first + count}]"
    return out


def compare_pair_fast(slots, omatches):
    """compare_pair without per-match Python work: phase-A slots [Ms, K] of one directed pair against the oracle's
    matches of that pair.  Returns dict(n_gpu, n_cpu, set_diff, inexact_fields, order_rows, tie_rows)."""
    Ms, K = slots.shape
    valid = slots["tgt_seg"] != EMPTY
    nv = valid.sum(1)
    assert np.array_equal(valid, np.arange(K)[None, :] < nv[:, None]), "valid slots are not a prefix of their row"
    rows = np.repeat(np.arange(Ms, dtype=np.uint64), K).reshape(Ms, K)[valid]
    gs = slots[valid]
    gk = (rows << np.uint64(32)) | gs["tgt_seg"].astype(np.uint64)
    ok_ = (omatches["src_seg"].astype(np.uint64) << np.uint64(32)) | omatches["tgt_seg"].astype(np.uint64)
    r = dict(n_gpu=len(gk), n_cpu=len(ok_), set_diff=0, inexact_fields=0, order_rows=0, tie_rows=0)
    if len(gk) == len(ok_) and np.array_equal(gk, ok_):
        go, oo = gs, omatches
    else:
        gi, oi = np.argsort(gk, kind="stable"), np.argsort(ok_, kind="stable")
        if len(gk) != len(ok_) or not np.array_equal(gk[gi], ok_[oi]):
            r["set_diff"] = len(np.setxor1d(gk, ok_))
            return r
        # same set: rows whose order differs, and whether each of them is an exact-overlap tie
        bad = np.unique(rows[gk != ok_])
        r["order_rows"] = len(bad)
        for s in bad:
            a = gs[rows == s]; b = omatches[omatches["src_seg"] == s]
            d = np.nonzero(a["tgt_seg"] != b["tgt_seg"])[0]
            r["tie_rows"] += int(np.array_equal(a["overlap"][d], b["overlap"][d]))
        go, oo = gs[gi], omatches[oi]
    for f in ("overlap", "d_p1", "d_p2", "d_q1", "d_q2"):
        r["inexact_fields"] += int(np.count_nonzero(go[f] != oo[f]))
    return r
