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
