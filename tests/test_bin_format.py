"""The BIN result format of Line3D::save3DLinesAsBIN (line3D.cc:2690-2711): boost::archive::binary_oarchive of
std::vector<FinalLine3D>, read and written without Boost (line3dpp_amd/io.py; the library's own writer is
l3d_save_3d_lines_bin).  Pinned on the reference's own fixtures testdata/Line3D++_ref/*vis_3.bin: parse -> re-serialise
is byte-identical for both files (where /root/reference exists), and a committed excerpt (first 40 records) agrees
with the TXT fixture of the same run to all six printed digits."""
import os

import numpy as np
import pytest

from line3dpp_amd import io

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF_DIR = "/root/reference/testdata/Line3D++_ref"


def test_excerpt_of_the_reference_fixture_parses_and_round_trips():
    raw = open(os.path.join(GOLD, "ref_lines3d_excerpt.bin"), "rb").read()
    lines, version = io.read_3d_lines_bin(os.path.join(GOLD, "ref_lines3d_excerpt.bin"))
    assert version == 10 and len(lines) == 40
    assert io.format_3d_lines_bin(lines, version) == raw
    txt = io.read_3d_lines_txt(os.path.join(GOLD, "ref_lines3d_excerpt.txt"))
    text = open(os.path.join(GOLD, "ref_lines3d_excerpt.txt")).read().split("\n")
    for L, T, row in zip(lines, txt, text):
        assert np.array_equal(L["residuals"], T["residuals"])
        assert L["segments"].shape == (len(T["segments"]), 9)
        # the TXT holds the same doubles printed with 6 significant digits (operator<<): identical text
        tok = row.split()
        printed = [io._g(v) for s in L["segments"] for v in s[:6]]
        assert tok[1:1 + len(printed)] == printed
        for g, ln, va in zip(L["segments"], L["seg_length"], L["seg_valid"]):
            assert va == 1 and abs(np.linalg.norm(g[3:6] - g[0:3]) - ln) < 1e-6 * max(ln, 1.0)
            assert abs(np.linalg.norm(g[6:9]) - 1.0) < 1e-9
        assert L["reference_view"] in set(L["residuals"][:, 0].tolist())


@pytest.mark.skipif(not os.path.isdir(REF_DIR), reason="reference checkout not present")
@pytest.mark.parametrize("tag", ["kNN_10__vis_3", "kNN_10__OPTIMIZED__vis_3"])
def test_full_reference_fixtures_round_trip_byte_for_byte(tag):
    path = os.path.join(REF_DIR, f"Line3D++__W_FULL__N_10__sigmaP_2.5__sigmaA_10__epiOverlap_0.25__{tag}.bin")
    lines, version = io.read_3d_lines_bin(path)
    assert io.format_3d_lines_bin(lines, version) == open(path, "rb").read()
    txt = io.read_3d_lines_txt(path[:-4] + ".txt")
    assert len(lines) == len(txt) >= 2489
    for L, T in zip(lines, txt):
        assert np.array_equal(L["residuals"], T["residuals"])
        assert np.allclose(L["segments"][:, :6], T["segments"], rtol=1e-5, atol=0)
