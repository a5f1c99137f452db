"""Input side: the segment cache of Line3D::detectLineSegments (boost binary archive of
DataArray<float4>, dataArray.h:352-374) and the VisualSfM .nvm reader of main_vsfm.cpp:144-250 (line3dpp_amd/io.py).
The reference ships neither a cache file nor its vsfm_result.nvm (missing blob), so the byte layout is pinned on the
archive rules that the reference's own BIN fixtures confirm (tests/test_bin_format.py: class header once, u64 counts) and
on hand-built files; the camera arithmetic is pinned on the reference's formulas."""
import struct

import numpy as np
import pytest

from line3dpp_amd import io


@pytest.mark.parametrize("n", [0, 1, 2, 7, 3000])
def test_segment_cache_round_trip_and_layout(tmp_path, n):
    rng = np.random.default_rng(n)
    segs = rng.uniform(0, 3000, (n, 4)).astype(np.float32)
    raw = io.format_segment_cache(segs)
    p = tmp_path / io.segment_cache_name(5, 3072, 2304)
    p.write_bytes(raw)
    assert p.name == "segments_L3D++_5_3072x2304_3000.bin"          # line3D.cc:300
    back = io.read_segment_cache(p)
    assert back.dtype == np.float32 and np.array_equal(back, segs)
    # layout: archive header (40 B), DataArray class header (5 B), 3 x u32 + 4 x u64, float4 class header, elements
    real = n + (n % 2)                                               # host rows padded to 32 B (dataArray.h:111-122)
    assert len(raw) == 40 + 5 + 12 + 32 + (5 if real else 0) + 16 * real
    w, h, rw, pitch, stride, pg, sg = struct.unpack_from("<IIIQQQQ", raw, 45)
    assert (w, h, rw, pitch, stride, pg, sg) == (n, 1, real, 16 * real, real, 0, 0)


def test_segment_cache_rejects_other_archives(tmp_path):
    p = tmp_path / "x.bin"
    p.write_bytes(b"not an archive at all, but long enough to be read as one........")
    with pytest.raises(ValueError):
        io.read_segment_cache(p)


def _write_nvm(path, cams, points):
    out = ["NVM_V3", "", str(len(cams))]
    for c in cams:
        out.append("%s %.10g %.12g %.12g %.12g %.12g %.12g %.12g %.12g %.6g 0" % (c["filename"], c["focal"], *c["q"], *c["C"], c["distortion"]))
    out += ["", str(len(points))]
    for p, meas in points:
        out.append("%.12g %.12g %.12g 128 128 128 %d " % (*p, len(meas)) + " ".join("%d %d %.3f %.3f" % m for m in meas))
    out += ["", "0", ""]
    path.write_text("\n".join(out))


def test_nvm_reader_follows_main_vsfm(tmp_path):
    rng = np.random.default_rng(3)
    cams = []
    for i in range(4):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        cams.append(dict(filename=f"img_{i}.jpg", focal=2400.0 + i, q=q, C=rng.normal(size=3) * 5, distortion=0.01 * i))
    points = []
    for k in range(30):
        p = rng.normal(size=3) * 3
        seen = sorted(rng.choice(4, size=rng.integers(2, 4), replace=False).tolist())
        points.append((p, [(c, k, 100.0 + k, 50.0) for c in seen]))
    path = tmp_path / "vsfm_result.nvm"
    _write_nvm(path, cams, points)
    got = io.read_nvm(path)
    assert len(got) == 4
    for i, (g, c) in enumerate(zip(got, cams)):
        qw, qx, qy, qz = c["q"]
        R = io.nvm_rotation(qw, qx, qy, qz)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-9) and np.isclose(np.linalg.det(R), 1.0)
        assert g["filename"] == c["filename"] and np.allclose(g["R"], R, atol=1e-9)
        assert np.allclose(g["t"], -R @ c["C"], atol=1e-8) and np.allclose(-g["R"].T @ g["t"], c["C"], atol=1e-8)
        ids = [k for k, (p, meas) in enumerate(points) if any(m[0] == i for m in meas)]
        assert g["worldpoints"] == ids
        d = sorted(np.float32(np.linalg.norm(points[k][0] - c["C"])) for k in ids)
        assert g["median_depth"] == d[len(d) // 2]
    K = io.nvm_intrinsics(got[0]["focal"], 3072, 2304)
    assert K[0, 2] == 1536.0 and K[1, 2] == 1152.0 and K[0, 0] == K[1, 1] == np.float32(2400.0)


def test_nvm_without_cameras_is_refused(tmp_path):
    p = tmp_path / "empty.nvm"
    p.write_text("NVM_V3\n\n0\n\n0\n")
    with pytest.raises(ValueError, match="No aligned cameras"):
        io.read_nvm(p)


# ---- the same readers behind the C-ABI (line3dpp_amd/csrc/l3d_io.hip): what a C++ caller of the library uses ----------
import ctypes as C  # noqa: E402


class _NvmCamera(C.Structure):
    _fields_ = [("filename", C.c_char_p), ("focal", C.c_float), ("distortion", C.c_float), ("median_depth", C.c_float),
                ("n_worldpoints", C.c_uint32), ("R", C.c_double * 9), ("t", C.c_double * 3), ("C", C.c_double * 3)]


def _lib():
    from line3dpp_amd import _lib as L
    return L, L.load()


@pytest.mark.parametrize("n", [0, 1, 2, 7, 3000])
def test_c_abi_segment_cache_equals_the_python_twin(tmp_path, n):
    L, lib = _lib()
    rng = np.random.default_rng(100 + n)
    segs = rng.uniform(0, 3000, (n, 4)).astype(np.float32)
    name = C.create_string_buffer(64)
    assert lib.l3d_segment_cache_name(5, 3072, 2304, 3000, name, 64) == 0
    assert name.value.decode() == io.segment_cache_name(5, 3072, 2304) == "segments_L3D++_5_3072x2304_3000.bin"
    assert lib.l3d_segment_cache_name(5, 3072, 2304, 3000, name, 8) != 0
    p = tmp_path / name.value.decode()
    assert lib.l3d_write_segment_cache(str(p).encode(), L.ptr(segs) if n else None, n) == 0
    assert p.read_bytes() == io.format_segment_cache(segs)                       # byte for byte what the twin writes
    assert np.array_equal(io.read_segment_cache(p), segs)
    cnt = C.c_uint32(99)
    assert lib.l3d_read_segment_cache(str(p).encode(), None, 0, C.byref(cnt)) == 0 and cnt.value == n
    back = np.zeros((max(n, 1), 4), np.float32)
    assert lib.l3d_read_segment_cache(str(p).encode(), L.ptr(back), n, C.byref(cnt)) == 0
    assert np.array_equal(back[:n], segs)
    if n > 1:
        assert lib.l3d_read_segment_cache(str(p).encode(), L.ptr(back), n - 1, C.byref(cnt)) == -9     # L3D_ERR_LIMIT


def test_c_abi_segment_cache_rejects_what_the_twin_rejects(tmp_path):
    L, lib = _lib()
    cnt = C.c_uint32(0)
    good = io.format_segment_cache(np.ones((3, 4), np.float32))
    cases = {"junk": b"not an archive at all, but long enough to be read as one........", "short": good[:50],
             "long": good + b"\0", "two_rows": good[:49] + struct.pack("<I", 2) + good[53:],
             "sizes": good[:32] + bytes([4, 4, 4, 8, 1, 0, 0, 0]) + good[40:]}
    for name, raw in cases.items():
        p = tmp_path / (name + ".bin")
        p.write_bytes(raw)
        with pytest.raises(ValueError):
            io.read_segment_cache(p)
        assert lib.l3d_read_segment_cache(str(p).encode(), None, 0, C.byref(cnt)) != 0, name
    assert lib.l3d_read_segment_cache(str(tmp_path / "missing.bin").encode(), None, 0, C.byref(cnt)) != 0


def test_c_abi_nvm_reader_equals_the_python_twin(tmp_path):
    L, lib = _lib()
    rng = np.random.default_rng(8)
    cams = []
    for i in range(6):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        cams.append(dict(filename=f"dir/img_{i}.jpg", focal=1800.0 + 37.25 * i, q=q, C=rng.normal(size=3) * 5, distortion=0.0125 * i))
    points = []
    for k in range(200):
        seen = sorted(rng.choice(5, size=rng.integers(2, 5), replace=False).tolist())     # camera 5 sees nothing
        points.append((rng.normal(size=3) * 3, [(c, k, 100.0 + k, 50.0) for c in seen]))
    path = tmp_path / "vsfm_result.nvm"
    _write_nvm(path, cams, points)
    want = io.read_nvm(path)
    h = C.c_void_p()
    assert lib.l3d_nvm_open(str(path).encode(), C.byref(h)) == 0 and lib.l3d_nvm_num_cameras(h) == 6
    for i, w in enumerate(want):
        cam = _NvmCamera()
        assert lib.l3d_nvm_get_camera(h, i, C.byref(cam)) == 0
        assert cam.filename.decode() == w["filename"] and cam.focal == w["focal"] and cam.distortion == w["distortion"]
        assert np.array_equal(np.array(cam.R).reshape(3, 3), w["R"]) and np.array_equal(np.array(cam.C), w["C"])
        assert np.allclose(np.array(cam.t), w["t"], rtol=0, atol=1e-13)
        assert cam.n_worldpoints == len(w["worldpoints"])
        ids = np.zeros(max(cam.n_worldpoints, 1), np.uint32)
        assert lib.l3d_nvm_get_worldpoints(h, i, L.ptr(ids), cam.n_worldpoints) == 0
        assert ids[:cam.n_worldpoints].tolist() == w["worldpoints"]
        if w["median_depth"] is None:
            assert cam.n_worldpoints == 0 and cam.median_depth == 0.0 and i == 5
        else:
            assert abs(cam.median_depth - w["median_depth"]) <= 2e-7 * w["median_depth"]
    assert lib.l3d_nvm_get_camera(h, 6, C.byref(_NvmCamera())) != 0
    lib.l3d_nvm_close(h)
    K = np.zeros(9)
    lib.l3d_nvm_intrinsics(C.c_float(want[0]["focal"]), 3073, 2305, L.ptr(K))
    assert np.array_equal(K.reshape(3, 3), io.nvm_intrinsics(want[0]["focal"], 3073, 2305))
    empty = tmp_path / "empty.nvm"
    empty.write_text("NVM_V3\n\n0\n\n0\n")
    assert lib.l3d_nvm_open(str(empty).encode(), C.byref(h)) == -6 and b"No aligned cameras" in lib.l3d_last_error()
    assert lib.l3d_nvm_open(str(tmp_path / "missing.nvm").encode(), C.byref(h)) != 0


def test_nvm_to_neighbours_end_to_end_on_the_host(tmp_path):
    """what main_vsfm.cpp does before matchImages, through the C-ABI alone: read the .nvm, hand every camera's
    worldpoint list over, get the visual neighbours -- equal to the reference's own code on the same cameras"""
    from line3dpp_amd.api import neighbors_from_worldpoints
    from line3dpp_amd.scene import add_worldpoints, make_scene
    from oracle import oracle as orc
    assert orc.have_reference()
    sc = make_scene(10, 20, n_neighbors=2, seed=77)
    X = add_worldpoints(sc, n_points=1200, seed=2, keep=0.3)
    cams = []
    for v in sc.views:
        # quaternion of R (w x y z), centre; the scene's cameras become NVM cameras
        R = v.R; w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
        q = np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
        cams.append(dict(filename=f"{v.cam}.jpg", focal=float(v.K[0, 0]), q=q, C=-R.T @ v.t, distortion=0.0))
    seen = {}
    for v in sc.views:
        for wp in v.worldpoints:
            seen.setdefault(wp, []).append(v.cam)
    order = sorted(seen)                                   # NVM numbers the points by their line
    points = [(X[wp], [(c, 0, 1.0, 1.0) for c in seen[wp]]) for wp in order]
    path = tmp_path / "scene.nvm"
    _write_nvm(path, cams, points)
    L, lib = _lib()
    h = C.c_void_p()
    assert lib.l3d_nvm_open(str(path).encode(), C.byref(h)) == 0
    ids, Ks, Rs, ts, wps = [], [], [], [], []
    for i in range(lib.l3d_nvm_num_cameras(h)):
        cam = _NvmCamera(); lib.l3d_nvm_get_camera(h, i, C.byref(cam))
        w = np.zeros(cam.n_worldpoints, np.uint32); lib.l3d_nvm_get_worldpoints(h, i, L.ptr(w), cam.n_worldpoints)
        K = np.zeros(9); lib.l3d_nvm_intrinsics(cam.focal, sc.views[i].width, sc.views[i].height, L.ptr(K))
        ids.append(i); Ks.append(K.reshape(3, 3)); Rs.append(np.array(cam.R).reshape(3, 3)); ts.append(np.array(cam.t)); wps.append(w.tolist())
    lib.l3d_nvm_close(h)
    got = neighbors_from_worldpoints(ids, Ks, Rs, ts, wps, 4)
    o = orc.Oracle(reference=True, by_worldpoints=True, threads=2)
    for i, v in enumerate(sc.views):
        assert o.add_view(i, v.segs, Ks[i], Rs[i], ts[i], v.width, v.height, 5.0, wps[i]) == 0
    o.match_images(num_neighbors=4, kNN=2)
    for i in ids:
        assert np.array_equal(got[i], o.visual_neighbors(i)) and len(got[i]) > 0


def test_c_abi_nvm_reader_tolerates_what_the_stream_parser_tolerates(tmp_path):
    """main_vsfm.cpp reads with getline + operator>>: extra blanks, a missing trailing section and a file that ends early
    are not errors there (the remaining points are simply not seen); a measurement that names a camera beyond the camera
    list would index out of bounds in the reference -- the library refuses it"""
    L, lib = _lib()
    h = C.c_void_p()
    p = tmp_path / "ragged.nvm"
    p.write_text("NVM_V3\n\n2\n   a.jpg   1000   1 0 0 0   0 0 0   0 0\nb.jpg 1200 1 0 0 0 1 0 0 0.5 0\n\n3\n"
                 "0 0 5 1 2 3 2 0 0 1 1 1 0 2 2\n0 1 6 1 2 3 1 1 7 3 3\n")          # third point missing: file ends early
    assert lib.l3d_nvm_open(str(p).encode(), C.byref(h)) == 0 and lib.l3d_nvm_num_cameras(h) == 2
    cam = _NvmCamera()
    assert lib.l3d_nvm_get_camera(h, 0, C.byref(cam)) == 0 and cam.n_worldpoints == 1 and cam.focal == 1000.0
    assert lib.l3d_nvm_get_camera(h, 1, C.byref(cam)) == 0 and cam.n_worldpoints == 2 and cam.distortion == 0.5
    assert np.array_equal(np.array(cam.R).reshape(3, 3), np.eye(3)) and list(cam.t) == [-1.0, 0.0, 0.0]
    ids = np.zeros(2, np.uint32)
    assert lib.l3d_nvm_get_worldpoints(h, 1, L.ptr(ids), 2) == 0 and ids.tolist() == [0, 1]
    assert lib.l3d_nvm_get_worldpoints(h, 1, L.ptr(ids), 1) == -9                     # L3D_ERR_LIMIT
    lib.l3d_nvm_close(h)
    got = io.read_nvm(p)
    assert [len(c["worldpoints"]) for c in got] == [1, 2]
    bad = tmp_path / "bad.nvm"
    bad.write_text("NVM_V3\n\n1\na.jpg 1000 1 0 0 0 0 0 0 0 0\n\n1\n0 0 5 1 2 3 1 4 0 1 1\n")   # camera 4 of 1
    assert lib.l3d_nvm_open(str(bad).encode(), C.byref(h)) != 0 and b"malformed" in lib.l3d_last_error()


# ---- COLMAP text results and bundler files (round 4: main_colmap.cpp:136-348, main_bundler.cpp:147-252) ----------------
def _write_colmap(folder, cams, images, points, comments=True):
    folder.mkdir(exist_ok=True)
    with open(folder / "cameras.txt", "w") as f:
        if comments:
            f.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n")
        for cid, model, w, h, params in cams:
            f.write(f"{cid} {model} {w} {h} " + " ".join(repr(float(p)) for p in params) + "\n")
    with open(folder / "images.txt", "w") as f:
        if comments:
            f.write("# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n")
        for iid, q, t, cid, name, pts in images:
            f.write(f"{iid} " + " ".join(repr(float(x)) for x in list(q) + list(t)) + f" {cid} {name}\n")
            f.write(" ".join(f"{x!r} {y!r} {pid}" for x, y, pid in pts) + "\n")
    with open(folder / "points3D.txt", "w") as f:
        if comments:
            f.write("# 3D point list with one line of data per point:\n#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[]\n")
        for pid, X in points:
            f.write(f"{pid} " + " ".join(repr(float(x)) for x in X) + " 128 128 128 0.5 1 2 3 4\n")


def _colmap_scene(rng, n_img=7):
    cams = [(1, "SIMPLE_PINHOLE", 3072, 2304, [2500.0, 1536.0, 1152.0]), (2, "PINHOLE", 3000, 2000, [2400.0, 2410.0, 1500.5, 999.5]),
            (3, "SIMPLE_RADIAL", 3072, 2304, [2500.0, 1536.0, 1152.0, -0.05]), (4, "RADIAL", 3072, 2304, [2500.0, 1536.0, 1152.0, -0.05, 0.01]),
            (5, "OPENCV", 3072, 2304, [2500.0, 2501.0, 1536.0, 1152.0, -0.05, 0.01, 1e-3, -2e-3]),
            (7, "FULL_OPENCV", 3072, 2304, [2500.0, 2501.0, 1536.0, 1152.0, -0.05, 0.01, 1e-3, -2e-3, 3e-4, 0.0, 0.0, 0.0])]
    points = [(100 + 3 * k, rng.normal(size=3) * 4) for k in range(150)]
    images = []
    for i in range(n_img):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        if i == 2:
            q *= 3.0                                     # not normalised: rotationFromQ divides by |q|^2
        cid = [1, 2, 3, 4, 5, 7, 9][i % 7]               # camera 9 does not exist: the image is dropped
        pts = []
        for k in rng.choice(len(points), size=0 if i == 4 else 40, replace=False):
            pts.append((float(rng.uniform(0, 3000)), float(rng.uniform(0, 2000)), points[k][0]))
        pts += [(1.0, 2.0, -1), (3.0, 4.0, 99999)]       # an unmatched feature; a track that points3D.txt does not hold
        images.append((10 + 5 * i, q, rng.normal(size=3) * 6, cid, f"sub/img {i}.jpg".replace(" ", "_"), pts if i != 4 else [(5.0, 6.0, -1)]))
    return cams, images, points


def test_colmap_reader_follows_main_colmap(tmp_path):
    rng = np.random.default_rng(4)
    cams, images, points = _colmap_scene(rng)
    _write_colmap(tmp_path / "sfm", cams, images, points)
    got = io.read_colmap(str(tmp_path / "sfm"))
    assert [g["id"] for g in got] == [10, 15, 20, 25, 30, 35]          # file order; the image of camera 9 is gone
    pts = dict(points)
    for g, (iid, q, t, cid, name, pl) in zip(got, images):
        assert g["id"] == iid and g["camera"] == cid and g["name"] == name
        R = g["R"]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1.0) < 1e-12
        qn = q / np.linalg.norm(q)                                      # the textbook matrix of the normalised quaternion
        w, x, y, z = qn
        assert np.allclose(R, [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                               [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                               [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], atol=1e-12)
        assert np.allclose(R @ g["C"] + g["t"], 0, atol=1e-12)           # C = R^T (-t)
        want_wps = [p for _, _, p in pl if p >= 0]
        assert g["worldpoints"] == want_wps
        if want_wps:
            d = sorted(np.float32(np.linalg.norm(g["C"] - pts.get(p, np.zeros(3)))) for p in want_wps)
            assert g["median_depth"] == d[len(d) // 2]
        else:
            assert g["median_depth"] is None and iid == 30
    k = {g["camera"]: g for g in got}
    assert np.array_equal(k[1]["K"], [[2500, 0, 1536], [0, 2500, 1152], [0, 0, 1]]) and not k[1]["radial"].any()
    assert np.array_equal(k[2]["K"], [[2400, 0, 1500.5], [0, 2410, 999.5], [0, 0, 1]])
    assert np.array_equal(k[3]["radial"], [-0.05, 0, 0]) and np.array_equal(k[4]["radial"], [-0.05, 0.01, 0])
    assert np.array_equal(k[5]["radial"], [-0.05, 0.01, 0]) and np.array_equal(k[5]["tangential"], [1e-3, -2e-3])
    assert np.array_equal(k[7]["radial"], [-0.05, 0.01, 3e-4]) and np.array_equal(k[7]["tangential"], [1e-3, -2e-3])
    # an unknown camera model is an error (main_colmap.cpp:221-226)
    _write_colmap(tmp_path / "bad", [(1, "THIN_PRISM_FISHEYE", 100, 100, [1.0] * 12)], [], [])
    with pytest.raises(ValueError, match="unknown"):
        io.read_colmap(str(tmp_path / "bad"))


def _check_sfm_handle(L, lib, h, want, colmap):
    assert lib.l3d_sfm_num_images(h) == len(want)
    for i, w in enumerate(want):
        im = L.SfmImage()
        assert lib.l3d_sfm_get_image(h, i, C.byref(im)) == 0
        assert im.id == w["id"] and np.array_equal(np.array(im.R).reshape(3, 3), w["R"]) and np.array_equal(np.array(im.t), w["t"])
        assert np.allclose(np.array(im.C), w["C"], rtol=0, atol=1e-12)
        assert np.array_equal(np.array(im.radial), w["radial"])
        if colmap:
            assert im.camera == w["camera"] and im.name.decode() == w["name"] and (im.width, im.height) == (w["width"], w["height"])
            assert np.array_equal(np.array(im.K).reshape(3, 3), w["K"]) and np.array_equal(np.array(im.tangential), w["tangential"])
        else:
            assert im.focal == w["focal"] and im.camera == i and not np.array(im.K).any()
        assert im.n_worldpoints == len(w["worldpoints"])
        ids = np.zeros(max(im.n_worldpoints, 1), np.uint32)
        assert lib.l3d_sfm_get_worldpoints(h, i, L.ptr(ids), im.n_worldpoints) == 0
        assert ids[:im.n_worldpoints].tolist() == w["worldpoints"]
        if im.n_worldpoints:
            assert lib.l3d_sfm_get_worldpoints(h, i, L.ptr(ids), im.n_worldpoints - 1) != 0       # buffer too small
        if w["median_depth"] is None:
            assert im.n_worldpoints == 0 and im.median_depth == 0.0
        else:
            assert abs(im.median_depth - w["median_depth"]) <= 2e-7 * w["median_depth"]
    assert lib.l3d_sfm_get_image(h, len(want), C.byref(L.SfmImage())) != 0


def test_c_abi_colmap_reader_equals_the_python_twin(tmp_path):
    L, lib = _lib()
    rng = np.random.default_rng(5)
    for comments in (True, False):
        cams, images, points = _colmap_scene(rng, n_img=9)
        folder = tmp_path / f"sfm{int(comments)}"
        _write_colmap(folder, cams, images, points, comments)
        want = io.read_colmap(str(folder))
        h = C.c_void_p()
        assert lib.l3d_sfm_open_colmap(str(folder).encode(), C.byref(h)) == 0
        _check_sfm_handle(L, lib, h, want, True)
        lib.l3d_sfm_close(h)
    h = C.c_void_p()
    assert lib.l3d_sfm_open_colmap(str(tmp_path / "nowhere").encode(), C.byref(h)) != 0 and b"does not exist" in lib.l3d_last_error()
    _write_colmap(tmp_path / "bad", [(1, "THIN_PRISM_FISHEYE", 100, 100, [1.0] * 12)], [], [])
    assert lib.l3d_sfm_open_colmap(str(tmp_path / "bad").encode(), C.byref(h)) != 0 and b"unknown" in lib.l3d_last_error()
    # two quirks of main_colmap.cpp that both readers must share:
    # (1) a repeated IMAGE_ID -- the reference's maps are keyed by the id, so both entries of the image sequence end up
    #     with the LAST pose and the LAST worldpoint list
    cams, images, points = _colmap_scene(rng, n_img=5)
    iid0 = images[0][0]
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    images.append((iid0, q, rng.normal(size=3) * 6, 2, "again.jpg", [(7.0, 8.0, points[3][0]), (9.0, 10.0, points[5][0])]))
    folder = tmp_path / "dup"
    _write_colmap(folder, cams, images, points)
    want = io.read_colmap(str(folder))
    assert [w["id"] for w in want].count(iid0) == 2
    a, b = [w for w in want if w["id"] == iid0]
    assert a["name"] == b["name"] == "again.jpg" and a["worldpoints"] == b["worldpoints"] == [points[3][0], points[5][0]]
    assert np.array_equal(a["R"], b["R"]) and a["camera"] == b["camera"] == 2
    assert lib.l3d_sfm_open_colmap(str(folder).encode(), C.byref(h)) == 0
    _check_sfm_handle(L, lib, h, want, True)
    lib.l3d_sfm_close(h)
    # (2) a blank line in cameras.txt is parsed like any other line and fails on its empty model name, in both
    folder = tmp_path / "blank"
    _write_colmap(folder, cams, images[:2], points)
    txt = (folder / "cameras.txt").read_text().split("\n")
    (folder / "cameras.txt").write_text("\n".join(txt[:3] + [""] + txt[3:]))
    with pytest.raises(ValueError, match="unknown"):
        io.read_colmap(str(folder))
    assert lib.l3d_sfm_open_colmap(str(folder).encode(), C.byref(h)) != 0 and b"unknown" in lib.l3d_last_error()


def _write_bundler(path, cams, points):
    with open(path, "w") as f:
        f.write(f"# Bundle file v0.3\n{len(cams)} {len(points)}\n")
        for c in cams:
            f.write(f"{c['f']!r} {c['k1']!r} {c['k2']!r}\n")
            for r in c["R"]:
                f.write(" ".join(repr(float(x)) for x in r) + "\n")
            f.write(" ".join(repr(float(x)) for x in c["t"]) + "\n")
        for X, views in points:
            f.write(" ".join(repr(float(x)) for x in X) + "\n200 100 50\n")
            f.write(f"{len(views)} " + " ".join(f"{cam} {key} {x!r} {y!r}" for cam, key, x, y in views) + "\n")


def test_bundler_reader_follows_main_bundler_and_the_c_abi_equals_the_twin(tmp_path):
    L, lib = _lib()
    rng = np.random.default_rng(6)
    cams = []
    for i in range(5):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        cams.append(dict(f=900.0 + 11.5 * i, k1=-0.01 * i, k2=0.002 * i, R=io.rotation_from_q(*q), t=rng.normal(size=3) * 3))
    points = []
    for k in range(120):
        seen = sorted(rng.choice(4, size=rng.integers(2, 4), replace=False).tolist())     # camera 4 sees nothing
        points.append((rng.normal(size=3) * 5, [(c, 7 * k, 10.0 + k, -3.5) for c in seen]))
    path = tmp_path / "bundle.rd.out"
    _write_bundler(path, cams, points)
    want = io.read_bundler(str(path))
    flip = np.diag([1.0, -1.0, -1.0])
    for w, c in zip(want, cams):
        assert np.array_equal(w["R"], flip @ c["R"]) and np.array_equal(w["t"], flip @ c["t"])   # y / z rows negated (:192-212)
        assert np.allclose(w["R"] @ w["C"] + w["t"], 0, atol=1e-12)
        assert np.allclose(w["C"], -c["R"].T @ c["t"], atol=1e-12)                              # the centre does not change
        assert w["focal"] == np.float32(c["f"]) and w["radial"][0] == np.float32(c["k1"]) and w["radial"][2] == 0
    assert want[4]["median_depth"] is None and want[4]["worldpoints"] == []
    assert want[0]["worldpoints"] == [k for k, (_, v) in enumerate(points) if any(c == 0 for c, *_ in v)]
    h = C.c_void_p()
    assert lib.l3d_sfm_open_bundler(str(path).encode(), C.byref(h)) == 0
    _check_sfm_handle(L, lib, h, want, False)
    lib.l3d_sfm_close(h)
    empty = tmp_path / "empty.out"
    empty.write_text("# Bundle file v0.3\n0 0\n")
    assert lib.l3d_sfm_open_bundler(str(empty).encode(), C.byref(h)) == -6 and b"No cameras" in lib.l3d_last_error()
    with pytest.raises(ValueError, match="No cameras"):
        io.read_bundler(str(empty))
    assert lib.l3d_sfm_open_bundler(str(tmp_path / "missing.out").encode(), C.byref(h)) != 0


def test_colmap_to_neighbours_end_to_end_on_the_host(tmp_path):
    """what main_colmap.cpp does before matchImages, through the C-ABI alone: read the result folder, hand every image's
    worldpoint list over (neighbors_by_worldpoints), get the visual neighbours -- equal to the reference's own code
    (oracle/_ref constructed in worldpoint mode) on the same cameras"""
    from line3dpp_amd.api import neighbors_from_worldpoints
    from line3dpp_amd.scene import add_worldpoints, make_scene
    from oracle import oracle as orc
    assert orc.have_reference()
    sc = make_scene(10, 20, n_neighbors=2, seed=78)
    X = add_worldpoints(sc, n_points=1200, seed=3, keep=0.3)
    cams, images = [], []
    for v in sc.views:
        cams.append((v.cam + 1, "PINHOLE", v.width, v.height, [v.K[0, 0], v.K[1, 1], v.K[0, 2], v.K[1, 2]]))
        R = v.R; w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
        q = np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
        images.append((v.cam, q, v.t, v.cam + 1, f"img_{v.cam}.jpg", [(1.0, 1.0, int(wp)) for wp in v.worldpoints]))
    _write_colmap(tmp_path / "sfm", cams, images, [(int(k), X[k]) for k in range(len(X))])
    L, lib = _lib()
    h = C.c_void_p()
    assert lib.l3d_sfm_open_colmap(str(tmp_path / "sfm").encode(), C.byref(h)) == 0
    ids, Ks, Rs, ts, wps, meds = [], [], [], [], [], []
    for i in range(lib.l3d_sfm_num_images(h)):
        im = L.SfmImage(); assert lib.l3d_sfm_get_image(h, i, C.byref(im)) == 0
        w = np.zeros(im.n_worldpoints, np.uint32); assert lib.l3d_sfm_get_worldpoints(h, i, L.ptr(w), im.n_worldpoints) == 0
        ids.append(int(im.id)); Ks.append(np.array(im.K).reshape(3, 3)); Rs.append(np.array(im.R).reshape(3, 3))
        ts.append(np.array(im.t)); wps.append(w.tolist()); meds.append(float(im.median_depth))
    lib.l3d_sfm_close(h)
    assert ids == [v.cam for v in sc.views] and all(m > 0 for m in meds)
    got = neighbors_from_worldpoints(ids, Ks, Rs, ts, wps, 4)
    o = orc.Oracle(reference=True, by_worldpoints=True, threads=2)
    for i, v in enumerate(sc.views):
        assert o.add_view(ids[i], v.segs, Ks[i], Rs[i], ts[i], v.width, v.height, meds[i], wps[i]) == 0
    o.match_images(num_neighbors=4, kNN=2)
    for i in ids:
        assert np.array_equal(got[i], o.visual_neighbors(i)) and len(got[i]) > 0
