"""CPU pin of the kNN tie order: line3dpp_amd/csrc/l3d_heap.h must pop exactly what std::priority_queue pops
(libstdc++, the container Line3D::matchingCPU keeps its kNN candidates in, line3D.cc:931-1007)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_heap_emulation_equals_std_priority_queue(tmp_path):
    exe = str(tmp_path / "heap_order")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "heap_order.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "identical" in out.stdout


def test_division_free_depth_decision_equals_the_reference_arithmetic(tmp_path):
    """l3d_dev.h: depths_positive (stage 1 of the match kernel's candidate pipeline: the accept / reject decision of
    Line3D::triangulationDepths + line3D.cc:966-980 with one multiplication per depth, falling back to the division in
    the sliver where the two roundings could disagree) against exact_depths on 2.4 million cases incl. depths within a
    few ulp of L3D_EPS, degenerate denominators, NaN and inf -- and, since round 4, the FLOAT decision on 48-byte records
    (depths_positive32) on another 2.4 million cases incl. rays within 1e-5 .. 1e-8 of the other segment's plane,
    baselines within 1e-5 .. 1e-9 of a plane, camera centres six orders of magnitude apart, tiny baselines and NaN: whenever
    it reports `certain` it must equal exact_depths (uncertain candidates take the double-precision decision)."""
    exe = str(tmp_path / "depth_sign")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           os.path.join(ROOT, "tests", "cpp", "depth_sign.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "identical" in out.stdout and " 0 mismatches" in out.stdout
    assert "float decision" in out.stdout and "0 mismatches among the certain" in out.stdout


def test_division_free_orientation_decision_equals_the_reference_arithmetic(tmp_path):
    """l3d_dev.h: orientation_ok_fast (checkMatchOrientation's keep / drop decision, line3D.cc:811-858 with
    View::segmentQualityAngle and unprojectSegment, as c^2 <= cos^2 * |v|^2 with a fallback to the reference's own
    arithmetic near the thresholds and for near-zero-length segments) against orientation_ok on 2.5 million cases incl.
    directions within a few ulp of both thresholds."""
    exe = str(tmp_path / "orient_sign")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           os.path.join(ROOT, "tests", "cpp", "orient_sign.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "identical" in out.stdout and " 0 mismatches" in out.stdout


def test_prefilter_without_reciprocals_never_rejects_an_accepted_pair(tmp_path):
    """l3d_dev.h: prefilter_products (the fp32 pre-filter of k_match_pairs in its form without v_rcp_f32) against the
    reference's arithmetic (exact_overlap = Line3D::matchingCPU line3D.cc:919-958 + mutualOverlap :1086-1165): on 61
    million (pair, threshold) cases -- random, along the epipolar band, nearly parallel to the pencil, end points on an
    epipolar line, thresholds up to the pair's own overlap -- no pair with overlap > threshold is rejected."""
    exe = str(tmp_path / "prefilter_cover")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           os.path.join(ROOT, "tests", "cpp", "prefilter_cover.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "covered" in out.stdout and " 0 lost" in out.stdout
