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
