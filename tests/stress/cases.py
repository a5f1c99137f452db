"""The random cases of the stress scripts (stress_pipeline.py, stress_culling.py) and of tests/test_gpu_stress.py, which runs
a fixed, seeded handful of them inside pytest."""
import numpy as np

from line3dpp_amd.scene import make_scene


def pipeline_case(rng, max_views=16, max_segs=700):
    """-> (scene, dict of matchImages parameters): random ring geometry, neighbour count, kNN (15 %: keep-all), overlap
    threshold, angle / position regulariser (negative: metric), ragged views and asymmetric neighbour lists in half of them"""
    nv = int(rng.integers(3, max_views)); ns = int(rng.integers(40, max_segs)); nn = int(rng.integers(2, min(nv, 12)))
    knn = int(rng.choice([1, 5, 10, 25])); epi = float(rng.choice([0.1, 0.25, 0.5])); sa = float(rng.choice([5.0, 10.0, 20.0]))
    sp = float(rng.choice([1.0, 2.5, 5.0, -0.05, -0.2])); radius = float(rng.uniform(10, 50))   # < 0: metric regulariser
    if rng.random() < 0.15:
        knn = 0                                                                             # keep-all mode
    sc = make_scene(nv, ns, n_neighbors=nn, seed=int(rng.integers(1, 1 << 30)), radius=radius, noise_px=float(rng.uniform(0, 1.5)),
                    real_fraction=float(rng.uniform(0.3, 0.9)))
    if rng.random() < 0.5:    # ragged views and asymmetric neighbour lists
        for v in sc.views:
            v.segs = v.segs[:max(1, int(len(v.segs) * rng.uniform(0.3, 1.0)))].copy()
            if len(v.neighbors) > 1 and rng.random() < 0.5:
                v.neighbors = v.neighbors[:-1]
    return sc, dict(sigma_p=sp, sigma_a=sa, kNN=knn, epi_overlap=epi)


def culling_case(rng, max_views=14, max_segs=2500):
    """-> (scene, kNN, epipolar overlap): random ring geometry, 30 % with an anisotropic rescale of the images"""
    nv = int(rng.integers(3, max_views)); ns = int(rng.integers(50, max_segs)); nn = int(rng.integers(2, min(nv, 8)))
    radius = float(rng.uniform(8, 60)); knn = int(rng.choice([1, 3, 10, 20])); epi = float(rng.choice([0.1, 0.25, 0.5, 0.8]))
    sc = make_scene(nv, ns, n_neighbors=nn, seed=int(rng.integers(1, 1 << 30)), radius=radius, noise_px=float(rng.uniform(0, 2)))
    if rng.random() < 0.3:   # anisotropic rescale of the image
        sx, sy = rng.uniform(0.5, 2.0, 2)
        for v in sc.views:
            v.segs = (v.segs * np.array([sx, sy, sx, sy])).astype(np.float32); v.K = v.K.copy(); v.K[0] *= sx; v.K[1] *= sy
            v.width = int(v.width * sx); v.height = int(v.height * sy)
    return sc, knn, epi
