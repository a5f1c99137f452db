"""Full-size parity without re-running the reference: a RESULT RECORD of matchImages + computeAffinity that can be taken
from the HIP context (line3dpp_amd.Line3D) and from the oracle (oracle.Oracle driving the reference's own code) alike.

  exact part   everything the north_star bar wants identical: per view the surviving match lists IN ORDER (segment
               offsets, target camera / segment, overlap and the four depths bit for bit) and View::k; the keys and the
               chosen match of every best hypothesis; the unordered segment pairs of A_.  Stored as SHA-256 digests
               (tests/golden/full/<config>.json, committed, a few KB per config).
  float part   score3D of every surviving match, 3D end points / direction of every best hypothesis, affinity weights,
               view median depths, med_scene_depth_lines: compared at REL_TOL.  Stored as arrays in
               oracle/_ref/cache/full_<config>.npz (git-ignored like oracle/_ref itself, travels with the snapshot) and,
               so that the float half of the check can never silently drop out, as a COMMITTED strided sample of the
               same arrays (tests/golden/full/<config>_floats_sample.npz: every n-th value, <= 16 384 per array, a few
               hundred KB per configuration).  load_reference() hands back the full arrays when they are there, the
               sample otherwise, and None only when neither exists -- which the callers treat as a FAILURE.

tools/ref_digest.py writes both from a run of oracle/_ref on the full BASELINE configuration (minutes of CPU, done once
per reference build); tests/test_gpu_full_size.py and bench.py --parity-digest compare the HIP result with them.
"""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIGEST_DIR = os.path.join(ROOT, "tests", "golden", "full")
CACHE_DIR = os.path.join(ROOT, "oracle", "_ref", "cache")


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def scene_hash(scene):
    h = hashlib.sha256()
    for v in scene.views:
        h.update(np.uint32(v.cam).tobytes()); h.update(np.ascontiguousarray(v.segs, np.float32).tobytes())
        h.update(np.ascontiguousarray(v.K, np.float64).tobytes()); h.update(np.ascontiguousarray(v.R, np.float64).tobytes())
        h.update(np.ascontiguousarray(v.t, np.float64).tobytes())
        h.update(np.asarray([v.width, v.height], np.uint32).tobytes()); h.update(np.float32(v.median_depth).tobytes())
        h.update(np.asarray(sorted(v.neighbors), np.uint32).tobytes())
    return h.hexdigest()


def result_record(x, scene, is_oracle):
    """(exact, floats): exact = dict of digests / counts, floats = dict of arrays"""
    views, score, med, n_surv = {}, [], [], 0
    for v in scene.views:
        m, off = x.matches(v.cam)
        info = x.view_info(v.cam)
        views[str(v.cam)] = _sha(np.asarray(off, np.uint32), m["tgt_cam"], m["tgt_seg"], m["src_seg"], m["overlap"],
                                 m["d_p1"], m["d_p2"], m["d_q1"], m["d_q2"], np.float32(info["k"]))
        score.append(np.asarray(m["score3D"], np.float32)); med.append(np.float32(info["median_depth"]))
        n_surv += len(m)
    if is_oracle:
        cs, geo, _, bm = x.best()
        key = (cs[:, 0].astype(np.uint64) << np.uint64(32)) | cs[:, 1].astype(np.uint64)
        ae, al = x.affinity()
        lk = (al[:, 0].astype(np.uint64) << np.uint64(32)) | al[:, 1].astype(np.uint64)
        msdl = np.float32(x.med_scene_depth_lines())
    else:
        s2, s3, bm = x.best()
        key = (s2["cam"].astype(np.uint64) << np.uint64(32)) | s2["seg"].astype(np.uint64)
        geo = np.concatenate([s3["P1"], s3["P2"], s3["dir"]], 1)
        ae, al, msdl = x.affinity()
        lk = (al["cam"].astype(np.uint64) << np.uint64(32)) | al["seg"].astype(np.uint64)
        msdl = np.float32(msdl)
    order = np.argsort(key, kind="stable")
    e = ae[0::2]                                              # A_ holds (i,j),(j,i) pairs
    a, b = lk[e["i"]], lk[e["j"]]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    eo = np.lexsort((hi, lo))
    exact = {"views": views, "surviving": int(n_surv), "best": int(len(key)), "affinity_entries": int(len(ae)),
             "best_digest": _sha(key[order], bm["tgt_cam"][order], bm["tgt_seg"][order]),
             "affinity_digest": _sha(lo[eo], hi[eo])}
    floats = {"score3D": np.concatenate(score) if score else np.zeros(0, np.float32), "median_depth": np.asarray(med, np.float32),
              "best_geo": np.asarray(geo, np.float64)[order], "affinity_w": np.asarray(e["w"], np.float32)[eo],
              "med_scene_depth_lines": np.asarray([msdl], np.float32)}
    return exact, floats


def _max_rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-30)))


SAMPLE_CAP = {"score3D": 16384, "affinity_w": 16384, "best_geo": 4096}   # values / rows kept of each array


def sample_floats(floats):
    """the committed compact form: every n-th value of the large arrays (the strides travel with it)"""
    out, strides = {}, {}
    for k, a in floats.items():
        a = np.asarray(a)
        st = max(1, -(-len(a) // SAMPLE_CAP[k])) if k in SAMPLE_CAP else 1
        out[k] = a[::st]; strides[k] = st
    out["_strides"] = np.asarray([strides[k] for k in FLOAT_KEYS], np.int64)
    return out


FLOAT_KEYS = ("score3D", "median_depth", "best_geo", "affinity_w", "med_scene_depth_lines")


def compare(exact, floats, ref_exact, ref_floats, tol):
    """-> dict(ok, differing_views, ..., max_rel); ref_floats: the full arrays or the committed sample (key "_strides");
    None leaves only the exact part checked, and r["floats_checked"] False -- callers must treat that as a failure"""
    r = {"surviving": ref_exact["surviving"], "best": ref_exact["best"], "affinity_entries": ref_exact["affinity_entries"]}
    r["differing_views"] = sorted(int(c) for c in ref_exact["views"] if exact["views"].get(c) != ref_exact["views"][c])
    r["counts_equal"] = all(exact[k] == ref_exact[k] for k in ("surviving", "best", "affinity_entries"))
    r["best_equal"] = exact["best_digest"] == ref_exact["best_digest"]
    r["affinity_pairs_equal"] = exact["affinity_digest"] == ref_exact["affinity_digest"]
    r["exact_ok"] = bool(not r["differing_views"] and r["counts_equal"] and r["best_equal"] and r["affinity_pairs_equal"])
    r["floats_checked"] = False if ref_floats is None else ("sample" if "_strides" in ref_floats else "full")
    r["max_rel"] = None
    if ref_floats is not None and r["exact_ok"]:
        if "_strides" in ref_floats:   # the candidate's arrays, sampled like the stored ones
            st = dict(zip(FLOAT_KEYS, (int(x) for x in ref_floats["_strides"])))
            floats = {k: np.asarray(floats[k])[::st[k]] for k in FLOAT_KEYS}
        rel = {k: _max_rel(floats[k], ref_floats[k]) for k in ("score3D", "median_depth", "affinity_w", "med_scene_depth_lines")}
        g, o = floats["best_geo"], np.asarray(ref_floats["best_geo"])
        err = 0.0
        for c in (0, 3, 6):
            if len(o):
                den = np.maximum(np.linalg.norm(o[:, c:c + 3], axis=1), 1e-30)
                err = max(err, float(np.max(np.linalg.norm(g[:, c:c + 3] - o[:, c:c + 3], axis=1) / den)))
        rel["best_geo"] = err
        r["max_rel_by_field"] = rel
        r["max_rel"] = max(rel.values())
    r["ok"] = bool(r["exact_ok"] and (r["max_rel"] is None or r["max_rel"] <= tol))
    return r


def load_reference(config):
    """(meta + exact digests, floats) of the stored reference run of `config`, or (None, None).  floats: the full arrays
    beside oracle/_ref when present, else the committed strided sample (dict with "_strides"), else None"""
    p = os.path.join(DIGEST_DIR, f"{config}.json")
    if not os.path.exists(p):
        return None, None
    meta = json.load(open(p))
    sha = _sha(json.dumps(meta["exact"], sort_keys=True).encode())
    floats = None
    fp = os.path.join(CACHE_DIR, f"full_{config}.npz")
    if os.path.exists(fp) and not os.environ.get("L3D_FLOATS_SAMPLE_ONLY"):
        z = np.load(fp)
        if str(z["exact_sha"]) == sha:
            floats = {k: z[k] for k in FLOAT_KEYS}
    if floats is None:
        sp = os.path.join(DIGEST_DIR, f"{config}_floats_sample.npz")
        if os.path.exists(sp):
            z = np.load(sp)
            if str(z["exact_sha"]) == sha:
                floats = {k: z[k] for k in FLOAT_KEYS + ("_strides",)}
    return meta, floats


def store_reference(config, meta, floats):
    os.makedirs(DIGEST_DIR, exist_ok=True); os.makedirs(CACHE_DIR, exist_ok=True)
    json.dump(meta, open(os.path.join(DIGEST_DIR, f"{config}.json"), "w"), indent=0, sort_keys=True)
    sha = _sha(json.dumps(meta["exact"], sort_keys=True).encode())
    np.savez_compressed(os.path.join(CACHE_DIR, f"full_{config}.npz"), exact_sha=sha, **floats)
    store_sample(config, sha, floats)


def store_sample(config, sha, floats):
    np.savez_compressed(os.path.join(DIGEST_DIR, f"{config}_floats_sample.npz"), exact_sha=sha,
                        **sample_floats({k: floats[k] for k in FLOAT_KEYS}))
