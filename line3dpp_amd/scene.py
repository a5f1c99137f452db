"""Synthetic N-view x M-segments-per-view scenes (BASELINE.json configs).

A box of planar facades carrying 3D line segments is observed by a ring of pinhole
cameras (3072x2304, f=2400 px, as the bundled testdata).  Per view the visible 3D
segments are projected, clipped, perturbed and padded with LSD-like clutter to exactly
M segments, sorted by length (line3D.cc:323-360 sorts detections the same way), and
handed over as explicit `line_segments` with explicit ring neighbours
(`neighbors_by_worldpoints=false`, main_mavmap.cpp:311-321).

Clutter statistics are those measured on the 17 603 2D segments of the reference
fixture testdata/Line3D++_ref/*.txt (log-normal length mu=3.91 sigma=0.60 px, ~62 %
within 10 deg of vertical, ~11 % within 10 deg of horizontal).
"""
from dataclasses import dataclass, field

import os

import numpy as np

WIDTH, HEIGHT, FOCAL = 3072, 2304, 2400.0
MIN_LEN = 0.005 * float(np.hypot(WIDTH, HEIGHT))  # commons.h:43 L3D_DEF_MIN_LINE_LENGTH_FACTOR


@dataclass
class ViewData:
    cam: int
    segs: np.ndarray          # [M,4] float32 (x1,y1,x2,y2) px
    K: np.ndarray             # [3,3] float64
    R: np.ndarray             # [3,3] float64   x = K [R|t] X
    t: np.ndarray             # [3]   float64
    width: int
    height: int
    median_depth: float
    neighbors: list
    worldpoints: list = field(default_factory=list)   # ids of the SfM points the view sees (neighbors_by_worldpoints)


@dataclass
class Scene:
    views: list = field(default_factory=list)
    name: str = ""

    @property
    def n_views(self):
        return len(self.views)

    def pair_tests(self):
        """Sum of Ms*Mt over the directed pairs matchImages visits (line3D.cc:704-741)."""
        ids = {v.cam: len(v.segs) for v in self.views}
        matched = {c: set() for c in ids}
        total = 0
        pairs = []
        for v in sorted(self.views, key=lambda v: v.cam):
            for n in sorted(set(v.neighbors)):
                if n in ids and n not in matched[v.cam]:
                    total += ids[v.cam] * ids[n]
                    pairs.append((v.cam, n))
                    matched[v.cam].add(n); matched[n].add(v.cam)
        return total, pairs


def _lookat(C, target):
    z = target - C; z /= np.linalg.norm(z)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(z, up); x /= np.linalg.norm(x)
    y = np.cross(z, x)          # image y points down
    return np.stack([x, y, z], 0)


def _clip_segments(p, q, w, h):
    """Liang-Barsky clip of 2D segments p->q ([n,2]) to [0,w-1]x[0,h-1]; returns mask, p', q'."""
    d = q - p
    t0 = np.zeros(len(p)); t1 = np.ones(len(p)); ok = np.ones(len(p), bool)
    for pk, qk in ((-d[:, 0], p[:, 0]), (d[:, 0], (w - 1) - p[:, 0]), (-d[:, 1], p[:, 1]), (d[:, 1], (h - 1) - p[:, 1])):
        par = np.abs(pk) < 1e-12
        ok &= ~(par & (qk < 0))
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(par, 0.0, qk / np.where(par, 1.0, pk))
        t0 = np.where(~par & (pk < 0), np.maximum(t0, r), t0)
        t1 = np.where(~par & (pk > 0), np.minimum(t1, r), t1)
    ok &= t0 < t1
    return ok, p + d * t0[:, None], p + d * t1[:, None]


def _clutter(rng, n):
    """n LSD-like 2D segments inside the image: [n,4] float64."""
    out = np.zeros((0, 4))
    while len(out) < n:
        m = int((n - len(out)) * 1.3) + 16
        length = np.clip(np.exp(rng.normal(3.91, 0.60, m)), MIN_LEN * 1.005, 1500.0)
        u = rng.random(m)
        ang = np.where(u < 0.62, np.pi / 2 + np.deg2rad(rng.uniform(-10, 10, m)),
                       np.where(u < 0.73, np.deg2rad(rng.uniform(-10, 10, m)), rng.uniform(0, np.pi, m)))
        c = np.stack([rng.uniform(0, WIDTH - 1, m), rng.uniform(0, HEIGHT - 1, m)], 1)
        hv = 0.5 * length[:, None] * np.stack([np.cos(ang), np.sin(ang)], 1)
        a, b = c - hv, c + hv
        ok = ((a >= 0).all(1) & (b >= 0).all(1) & (a[:, 0] <= WIDTH - 1) & (b[:, 0] <= WIDTH - 1) &
              (a[:, 1] <= HEIGHT - 1) & (b[:, 1] <= HEIGHT - 1))
        flip = rng.random(m) < 0.5
        a2 = np.where(flip[:, None], b, a); b2 = np.where(flip[:, None], a, b)
        out = np.concatenate([out, np.concatenate([a2, b2], 1)[ok]], 0)
    return out[:n]


def _scene_lines(rng, S, half=10.0):
    """S 3D segments on the 4 vertical facades + roof of a box; returns P,Q [S,3], normal [S,3]."""
    face = rng.integers(0, 5, S)
    length = np.clip(np.exp(rng.normal(0.0, 0.6, S)), 0.15, 8.0)
    u = rng.uniform(-half, half, S); v = rng.uniform(-half, half, S)
    r = rng.random(S)
    ang = np.where(r < 0.35, 0.0, np.where(r < 0.70, np.pi / 2, rng.uniform(0, np.pi, S)))
    du, dv = 0.5 * length * np.cos(ang), 0.5 * length * np.sin(ang)
    ua, va, ub, vb = u - du, v - dv, u + du, v + dv
    for arr in (ua, va, ub, vb):
        np.clip(arr, -half, half, out=arr)
    P = np.zeros((S, 3)); Q = np.zeros((S, 3)); N = np.zeros((S, 3))
    for f in range(5):
        m = face == f
        if f == 0:   # +x facade: (half, u, v)
            P[m] = np.stack([np.full(m.sum(), half), ua[m], va[m]], 1); Q[m] = np.stack([np.full(m.sum(), half), ub[m], vb[m]], 1); N[m] = [1, 0, 0]
        elif f == 1:  # -x
            P[m] = np.stack([np.full(m.sum(), -half), ua[m], va[m]], 1); Q[m] = np.stack([np.full(m.sum(), -half), ub[m], vb[m]], 1); N[m] = [-1, 0, 0]
        elif f == 2:  # +y
            P[m] = np.stack([ua[m], np.full(m.sum(), half), va[m]], 1); Q[m] = np.stack([ub[m], np.full(m.sum(), half), vb[m]], 1); N[m] = [0, 1, 0]
        elif f == 3:  # -y
            P[m] = np.stack([ua[m], np.full(m.sum(), -half), va[m]], 1); Q[m] = np.stack([ub[m], np.full(m.sum(), -half), vb[m]], 1); N[m] = [0, -1, 0]
        else:         # roof z=+half
            P[m] = np.stack([ua[m], va[m], np.full(m.sum(), half)], 1); Q[m] = np.stack([ub[m], vb[m], np.full(m.sum(), half)], 1); N[m] = [0, 0, 1]
    return P, Q, N


def make_scene(n_views, n_segs, n_neighbors=10, seed=0x4C334450, real_fraction=0.5, radius=25.0,
               noise_px=0.5, rings=1, name="", max_views=None):
    """Ring-of-cameras scene: `n_views` views x exactly `n_segs` segments, ring neighbours +-n/2.
    max_views: stop after the first `max_views` views (they are identical to the first views of the full scene: the
    random stream is consumed view by view) -- for tests that need a slice of a large configuration."""
    rng = np.random.default_rng(seed)
    S = max(int(6 * n_segs * real_fraction), 64)
    P, Q, N = _scene_lines(rng, S)
    K = np.array([[FOCAL, 0, WIDTH / 2], [0, FOCAL, HEIGHT / 2], [0, 0, 1.0]])
    views = []
    per_ring = n_views // rings
    for i in range(n_views if max_views is None else min(n_views, max_views)):
        ring, j = divmod(i, per_ring) if rings > 1 else (0, i)
        nr = per_ring if rings > 1 else n_views
        phi = 2 * np.pi * (j + 0.37 * ring) / nr
        rad = radius * (1.0 + 0.6 * ring)
        Cc = np.array([rad * np.cos(phi), rad * np.sin(phi), rng.uniform(-2, 2) + 6.0 * ring])
        R = _lookat(Cc, rng.normal(0, 0.5, 3))
        t = -R @ Cc
        # project the 3D segments facing the camera
        facing = ((Cc[None, :] - P) * N).sum(1) > 0.5
        Xc_p = (R @ P.T).T + t; Xc_q = (R @ Q.T).T + t
        front = (Xc_p[:, 2] > 1.0) & (Xc_q[:, 2] > 1.0) & facing
        xp = (K @ Xc_p.T).T; xq = (K @ Xc_q.T).T
        with np.errstate(divide="ignore", invalid="ignore"):
            p2 = xp[:, :2] / xp[:, 2:3]; q2 = xq[:, :2] / xq[:, 2:3]
        idx = np.nonzero(front)[0]
        ok, pc, qc = _clip_segments(p2[idx], q2[idx], WIDTH, HEIGHT)
        pc, qc = pc[ok], qc[ok]
        keep = rng.random(len(pc)) > 0.2
        pc, qc = pc[keep], qc[keep]
        pc = pc + rng.normal(0, noise_px, pc.shape); qc = qc + rng.normal(0, noise_px, qc.shape)
        inside = ((pc >= 0).all(1) & (qc >= 0).all(1) & (pc[:, 0] <= WIDTH - 1) & (qc[:, 0] <= WIDTH - 1) &
                  (pc[:, 1] <= HEIGHT - 1) & (qc[:, 1] <= HEIGHT - 1))
        long_enough = np.hypot(*(pc - qc).T) >= MIN_LEN
        real = np.concatenate([pc, qc], 1)[inside & long_enough]
        max_real = int(n_segs * real_fraction)
        if len(real) > max_real:
            real = real[rng.permutation(len(real))[:max_real]]
        segs = np.concatenate([real, _clutter(rng, n_segs - len(real))], 0)
        length = np.hypot(segs[:, 0] - segs[:, 2], segs[:, 1] - segs[:, 3])
        segs = segs[np.argsort(-length, kind="stable")].astype(np.float32)
        if rings > 1:
            base = ring * per_ring
            nb = [base + (j + d) % per_ring for d in range(-(n_neighbors // 2), n_neighbors // 2 + 1) if d != 0]
        else:
            nb = [(i + d) % n_views for d in range(-(n_neighbors // 2), n_neighbors // 2 + 1) if d != 0]
        nb = sorted(set(n for n in nb if n != i))
        views.append(ViewData(i, segs, K.copy(), R, t, WIDTH, HEIGHT, float(np.linalg.norm(Cc)), nb))
    return Scene(views, name or f"ring{n_views}x{n_segs}n{n_neighbors}")


# BASELINE.json configs.  C0 = the reference's bundled testdata (26 images): its SfM input vsfm_result.nvm is not
# part of the reference checkout, so the cameras were recovered from the reference's own result fixture by line-based
# resection and the 2D segments are those that occur in that result (tests/golden/make_real_scene.py).
C0_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "real_scene_c0.npz")
CONFIGS = {
    "C0": dict(n_views=26, n_segs=670, n_neighbors=10),   # n_segs: average; 406-919 per view
    "C1": dict(n_views=64, n_segs=2000, n_neighbors=10),
    "C2": dict(n_views=256, n_segs=4096, n_neighbors=20),
    "C3": dict(n_views=1024, n_segs=1000, n_neighbors=10, rings=2, radius=22.0),
    "C4": dict(n_views=128, n_segs=16384, n_neighbors=10),
}


def load_scene_npz(path, name=""):
    """scene stored by tests/golden/make_real_scene.py: cameras, per-view segments, neighbours, median depths"""
    d = np.load(path)
    views = []
    for i, cam in enumerate(d["cam"]):
        a, b = int(d["seg_off"][i]), int(d["seg_off"][i + 1])
        nb = [int(x) for x in d["nb"][int(d["nb_off"][i]):int(d["nb_off"][i + 1])]]
        views.append(ViewData(int(cam), d["segs"][a:b].astype(np.float32), d["K"][i].copy(), d["R"][i].copy(), d["t"][i].copy(),
                              int(d["width"]), int(d["height"]), float(d["median_depth"][i]), nb))
    return Scene(views, name or os.path.basename(path))


def make_config(name, seed=None, max_views=None):
    if name == "C0":
        return load_scene_npz(C0_FILE, "C0")
    idx = list(CONFIGS).index(name)
    return make_scene(seed=(0x4C334450 + idx) if seed is None else seed, name=name, max_views=max_views, **CONFIGS[name])


def add_worldpoints(scene, n_points=4000, seed=7, extent=12.0, keep=0.7):
    """SfM-like worldpoint lists for the views of `scene` (the input of Line3D instances constructed with
    neighbors_by_worldpoints=true): random 3D points around the origin; a view sees a point that lies in front of it,
    projects into its image and survives a random drop-out (a feature detector's miss).  Returns the points."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-extent, extent, (n_points, 3)) * np.array([1.0, 1.0, 0.4])
    for v in scene.views:
        Xc = (v.R @ X.T).T + v.t
        x = (v.K @ Xc.T).T
        with np.errstate(divide="ignore", invalid="ignore"):
            u = x[:, 0] / x[:, 2]; w = x[:, 1] / x[:, 2]
        vis = (Xc[:, 2] > 0.5) & (u >= 0) & (u < v.width) & (w >= 0) & (w < v.height) & (rng.random(n_points) < keep)
        v.worldpoints = [int(i) for i in np.nonzero(vis)[0]]
    return X
