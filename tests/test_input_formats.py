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
