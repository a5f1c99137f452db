"""The library's readers of the reference's three SfM input formats -- VisualSfM .nvm, COLMAP text models, bundler --
against the reference's OWN front ends: main_vsfm.cpp, main_colmap.cpp and main_bundler.cpp compiled in place
(oracle/Makefile: oracle/_ref/libl3d_ref_front.so) against a RECORDER of the Line3D interface (oracle/ref_shim_front).
Running a front end on an SfM result leaves the calls it makes on Line3D; what its parser hands to addImage -- camera id,
K, R, t, median depth, worldpoint ids, and the distortion it hands to undistortImage -- is what the readers behind the
C-ABI (l3d_nvm_*, l3d_sfm_* in line3dpp_amd/csrc/l3d_io.hip) and their Python twins (line3dpp_amd/io.py) must return.
rotationFromQ inside the front ends is the reference's own (oracle/_ref/libl3d_ref.so).

The reference ships no SfM result of any of the three kinds, so the FILES are generated here; the PARSERS are the
reference's."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from line3dpp_amd import io
from tests.test_input_formats import _colmap_scene, _lib, _write_bundler, _write_colmap, _write_nvm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRONT = os.path.join(ROOT, "oracle", "_ref", "libl3d_ref_front.so")
REF = os.path.join(ROOT, "oracle", "_ref", "libl3d_ref.so")


def _front():
    assert os.path.exists(FRONT) and os.path.exists(REF), \
        "oracle/_ref/libl3d_ref_front.so is missing: `make -C oracle` builds it where /root/reference is present"
    f, r = C.CDLL(FRONT), C.CDLL(REF)
    f.lo_front_log.restype = C.c_char_p
    f.lo_front_set_rotation(C.cast(r.lo_ref_rotation_from_q, C.c_void_p))
    return f


def _run(f, which, args):
    argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
    rc = f.lo_front_run(which.encode(), len(args), argv)
    return rc, json.loads(f.lo_front_log().decode())


def _touch(folder, names):
    for n in names:
        p = folder / n
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(b"")


def _m(e, key):
    return np.array(e[key], np.float64).reshape(3, 3)


def _calls(events, name):
    return [e for e in events if e["call"] == name]


def _tail_is_the_reference_pipeline(events, out_folder, **match):
    """after the images: matchImages with the command line's values, reconstruct3Dlines, get3Dlines and the four writers"""
    names = [e["call"] for e in events]
    k = names.index("matchImages")
    assert names[k:] == ["matchImages", "reconstruct3Dlines", "get3Dlines", "saveResultAsSTL", "saveResultAsOBJ",
                         "save3DLinesAsTXT", "save3DLinesAsBIN"]
    for key, v in match.items():
        assert events[k][key] == v, (key, events[k][key], v)
    assert all(e["folder"] == out_folder for e in events[k + 3:])


def test_colmap_reader_equals_what_main_colmap_hands_to_addImage(tmp_path):
    rng = np.random.default_rng(4)
    cams, images, points = _colmap_scene(rng)
    _write_colmap(tmp_path / "sfm", cams, images, points)
    _touch(tmp_path / "imgs", [im[4] for im in images])
    out = str(tmp_path / "out")
    rc, ev = _run(_front(), "colmap", ["-i", str(tmp_path / "imgs"), "-m", str(tmp_path / "sfm"), "-o", out, "-n", "7", "-k", "4"])
    assert rc == 0
    assert ev[0]["call"] == "Line3D" and ev[0]["output_folder"] == out and ev[0]["neighbors_by_worldpoints"] == 1
    added = _calls(ev, "addImage")
    # ---- the Python twin ----
    got = [g for g in io.read_colmap(str(tmp_path / "sfm")) if g["worldpoints"]]      # :389-410: no worldpoints, no addImage
    assert [a["camID"] for a in added] == [g["id"] for g in got] and len(got) == 5
    for a, g in zip(added, got):
        assert np.array_equal(_m(a, "K"), g["K"]) and np.array_equal(_m(a, "R"), g["R"]) and np.array_equal(a["t"], g["t"])
        assert a["wps"] == g["worldpoints"] and np.float32(a["median_depth"]) == g["median_depth"]
        assert (a["cols"], a["rows"]) == (640, 480) and a["n_segments"] == 0
    # distortion: one undistortImage per image whose camera has any (also the image without worldpoints, which is
    # undistorted before it is found to have none, :375-388), with the camera's coefficients and K
    und = _calls(ev, "undistortImage")
    dist = [g for g in io.read_colmap(str(tmp_path / "sfm")) if g["radial"].any() or g["tangential"].any()]
    assert len(und) == len(dist) == 4
    for u, g in zip(und, dist):
        assert np.array_equal(u["radial"], g["radial"]) and np.array_equal(u["tangential"], g["tangential"])
        assert np.array_equal(_m(u, "K"), g["K"])
    # ---- the C-ABI ----
    L, lib = _lib()
    h = C.c_void_p()
    assert lib.l3d_sfm_open_colmap(str(tmp_path / "sfm").encode(), C.byref(h)) == 0
    k = 0
    for i in range(lib.l3d_sfm_num_images(h)):
        im = L.SfmImage()
        assert lib.l3d_sfm_get_image(h, i, C.byref(im)) == 0
        if not im.n_worldpoints:
            continue
        a = added[k]; k += 1
        ids = np.zeros(im.n_worldpoints, np.uint32)
        assert lib.l3d_sfm_get_worldpoints(h, i, L.ptr(ids), im.n_worldpoints) == 0
        assert im.id == a["camID"] and ids.tolist() == a["wps"]
        assert np.array_equal(np.array(im.K).reshape(3, 3), _m(a, "K")) and np.array_equal(np.array(im.R).reshape(3, 3), _m(a, "R"))
        assert np.array_equal(np.array(im.t), a["t"]) and np.float32(im.median_depth) == np.float32(a["median_depth"])
    assert k == len(added)
    lib.l3d_sfm_close(h)
    _tail_is_the_reference_pipeline(ev, out, num_neighbors=7, kNN=4, sigma_position=2.5, sigma_angle=10.0)


def test_colmap_front_end_refuses_what_the_reader_refuses(tmp_path):
    f = _front()
    _write_colmap(tmp_path / "bad", [(1, "THIN_PRISM_FISHEYE", 100, 100, [1.0] * 12)], [], [])
    rc, ev = _run(f, "colmap", ["-i", str(tmp_path), "-m", str(tmp_path / "bad"), "-o", str(tmp_path / "o")])
    assert rc != 0 and not _calls(ev, "addImage")                      # main_colmap.cpp:221-226: unknown camera model
    with pytest.raises(ValueError, match="unknown"):
        io.read_colmap(str(tmp_path / "bad"))
    rc, ev = _run(f, "colmap", ["-i", str(tmp_path), "-m", str(tmp_path / "missing"), "-o", str(tmp_path / "o")])
    assert rc != 0 and not ev


def test_nvm_reader_equals_what_main_vsfm_hands_to_addImage(tmp_path):
    rng = np.random.default_rng(3)
    cams = []
    for i in range(5):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        cams.append(dict(filename=f"some/where/img_{i}.jpg", focal=2400.0 + 1.25 * i, q=q, C=rng.normal(size=3) * 5,
                         distortion=0.0 if i == 1 else 0.01 * (i + 1)))
    points = []
    for k in range(60):
        seen = sorted(rng.choice(4, size=rng.integers(2, 4), replace=False).tolist())   # camera 4 sees nothing: no addImage
        points.append((rng.normal(size=3) * 3, [(c, k, 100.0 + k, 50.0) for c in seen]))
    path = tmp_path / "vsfm_result.nvm"
    _write_nvm(path, cams, points)
    _touch(tmp_path / "imgs", [f"img_{i}.jpg" for i in range(5)])      # -i given: the file NAME of the .nvm path is used
    out = str(tmp_path / "out")
    rc, ev = _run(_front(), "vsfm", ["-i", str(tmp_path / "imgs"), "-m", str(path), "-o", out, "-e", "-0.4"])
    assert rc == 0
    added = _calls(ev, "addImage")
    got = io.read_nvm(path)
    assert [a["camID"] for a in added] == [0, 1, 2, 3]
    for a in added:
        g = got[a["camID"]]
        assert np.array_equal(_m(a, "R"), g["R"]) and np.array_equal(a["t"], g["t"])
        assert a["wps"] == g["worldpoints"] and np.float32(a["median_depth"]) == g["median_depth"]
        assert np.array_equal(_m(a, "K"), io.nvm_intrinsics(g["focal"], a["cols"], a["rows"]))      # :272-283
    und = _calls(ev, "undistortImage")
    assert len(und) == 3                                               # camera 1 has no distortion, camera 4 no points
    for u, i in zip(und, [0, 2, 3]):
        assert u["radial"] == [-float(got[i]["distortion"]), 0.0, 0.0] and u["tangential"] == [0.0, 0.0]   # :288-291
    # ---- the C-ABI ----
    from tests.test_input_formats import _NvmCamera
    L, lib = _lib()
    h = C.c_void_p()
    assert lib.l3d_nvm_open(str(path).encode(), C.byref(h)) == 0
    for a in added:
        c = _NvmCamera()
        assert lib.l3d_nvm_get_camera(h, a["camID"], C.byref(c)) == 0
        assert np.array_equal(np.array(c.R).reshape(3, 3), _m(a, "R")) and np.array_equal(np.array(c.t), a["t"])
        assert c.median_depth == np.float32(a["median_depth"]) and c.n_worldpoints == len(a["wps"])
        K = np.zeros(9)
        lib.l3d_nvm_intrinsics(c.focal, a["cols"], a["rows"], L.ptr(K))
        assert np.array_equal(K.reshape(3, 3), _m(a, "K"))
    lib.l3d_nvm_close(h)
    _tail_is_the_reference_pipeline(ev, out, epipolar_overlap=float(np.float32(0.4)))     # fmin(fabs(-0.4), 0.99), :128


def test_bundler_reader_equals_what_main_bundler_hands_to_addImage(tmp_path):
    rng = np.random.default_rng(6)
    cams = []
    for i in range(5):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        cams.append(dict(f=900.0 + 11.5 * i, k1=0.0 if i == 2 else -0.01 * (i + 1), k2=0.0 if i == 2 else 0.002 * i,
                         R=io.rotation_from_q(*q), t=rng.normal(size=3) * 3))
    points = []
    for k in range(120):
        seen = sorted(rng.choice(4, size=rng.integers(2, 4), replace=False).tolist())     # camera 4 sees nothing
        points.append((rng.normal(size=3) * 5, [(c, 7 * k, 10.0 + k, -3.5) for c in seen]))
    path = tmp_path / "bundle.rd.out"
    _write_bundler(path, cams, points)
    _touch(tmp_path / "imgs", [f"{i:08d}.jpg" for i in range(5)])      # main_bundler.cpp:296-316: zero-padded index + extension
    out = str(tmp_path / "out")
    rc, ev = _run(_front(), "bundler", ["-i", str(tmp_path / "imgs"), "-b", str(path), "-o", out])
    assert rc == 0
    added = _calls(ev, "addImage")
    got = io.read_bundler(str(path))
    assert [a["camID"] for a in added] == [0, 1, 2, 3]
    for a in added:
        g = got[a["camID"]]
        assert np.array_equal(_m(a, "R"), g["R"]) and np.array_equal(a["t"], g["t"])
        assert a["wps"] == g["worldpoints"] and np.float32(a["median_depth"]) == g["median_depth"]
        assert np.array_equal(_m(a, "K"), io.nvm_intrinsics(g["focal"], a["cols"], a["rows"]))      # :341-350
    und = _calls(ev, "undistortImage")
    assert len(und) == 3
    for u, i in zip(und, [0, 1, 3]):
        assert u["radial"] == [float(got[i]["radial"][0]), float(got[i]["radial"][1]), 0.0] and u["tangential"] == [0.0, 0.0]
    # ---- the C-ABI ----
    L, lib = _lib()
    h = C.c_void_p()
    assert lib.l3d_sfm_open_bundler(str(path).encode(), C.byref(h)) == 0
    for a in added:
        im = L.SfmImage()
        assert lib.l3d_sfm_get_image(h, a["camID"], C.byref(im)) == 0
        ids = np.zeros(im.n_worldpoints, np.uint32)
        assert lib.l3d_sfm_get_worldpoints(h, a["camID"], L.ptr(ids), im.n_worldpoints) == 0
        assert ids.tolist() == a["wps"] and np.float32(im.median_depth) == np.float32(a["median_depth"])
        assert np.array_equal(np.array(im.R).reshape(3, 3), _m(a, "R")) and np.array_equal(np.array(im.t), a["t"])
    lib.l3d_sfm_close(h)
    _tail_is_the_reference_pipeline(ev, out)
