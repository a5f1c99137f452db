#!/bin/bash
# round-3 GPU call 23: which warm-up removes the 8 ms first copy of a fresh process on C0 / C3 (runtime copies of several
# sizes issued through ctypes before the first call; the library is unchanged)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03w; mkdir -p $O; cd $R
cat > /tmp/first_call2.py <<'PY'
import ctypes as C, sys, time
sys.path.insert(0, sys.argv[3])
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
mode = sys.argv[2]
sc = make_config(sys.argv[1])
g = Line3D(); g.add_scene(sc)
if mode != "plain":
    hip = C.CDLL("libamdhip64.so")
    d = C.c_void_p(); h = C.c_void_p()
    N = 8 << 20
    assert hip.hipMalloc(C.byref(d), C.c_size_t(N)) == 0 and hip.hipHostMalloc(C.byref(h), C.c_size_t(N), 0) == 0
    sizes = {"sizes": [64, 1024, 4096, 16384, 65536, 262144, 1 << 20, 8 << 20], "many64k": [65536] * 16, "small": [64, 256, 1024, 4096, 8192, 16384, 32768]}[mode]
    t0 = time.time()
    for s in sizes:
        assert hip.hipMemcpyAsync(d, h, C.c_size_t(s), 1, None) == 0     # H2D
        assert hip.hipMemcpyAsync(h, d, C.c_size_t(s), 2, None) == 0     # D2H
    hip.hipDeviceSynchronize()
    print("warm-up copies (%s): %.2f ms" % (mode, (time.time() - t0) * 1e3))
ta=time.time(); ok = g.matchImages() and g.computeAffinity(); tb=time.time()
print("%s %s first call %.2f ms, begin %.3f" % (sys.argv[1], mode, (tb-ta)*1e3, g.timings()["begin_ms"]), flush=True)
PY
for c in C0 C3 C1; do for m in plain sizes many64k small; do python /tmp/first_call2.py $c $m $R 2>&1 | grep -v "^\[L3D" | tail -2; done; done
