#!/bin/bash
# round-3 GPU call 24: does the extra asynchronous copy help when issued BEFORE addImage (= could l3d_create do it)?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03x; mkdir -p $O; cd $R
cat > /tmp/first_call3.py <<'PY'
import ctypes as C, sys, time
sys.path.insert(0, sys.argv[3])
from line3dpp_amd.api import Line3D
from line3dpp_amd.scene import make_config
mode = sys.argv[2]
sc = make_config(sys.argv[1])
def copies(tag):
    hip = C.CDLL("libamdhip64.so")
    d = C.c_void_p(); h = C.c_void_p()
    N = 1 << 20
    assert hip.hipMalloc(C.byref(d), C.c_size_t(N)) == 0 and hip.hipHostMalloc(C.byref(h), C.c_size_t(N), 0) == 0
    t0 = time.time()
    for s in (64, 4096, 65536, 1 << 20):
        assert hip.hipMemcpyAsync(d, h, C.c_size_t(s), 1, None) == 0
        assert hip.hipMemcpyAsync(h, d, C.c_size_t(s), 2, None) == 0
    hip.hipDeviceSynchronize()
    print("copies %s: %.2f ms" % (tag, (time.time() - t0) * 1e3))
g = Line3D()
if mode == "before_add": copies("before addImage")
t0 = time.time(); g.add_scene(sc); print("add_scene %.1f ms" % ((time.time() - t0) * 1e3))
if mode == "after_add": copies("after addImage")
ta=time.time(); ok = g.matchImages() and g.computeAffinity(); tb=time.time()
print("%s %s first call %.2f ms, begin %.3f" % (sys.argv[1], mode, (tb-ta)*1e3, g.timings()["begin_ms"]), flush=True)
PY
for c in C0 C3; do for m in plain before_add after_add; do python /tmp/first_call3.py $c $m $R 2>&1 | grep -v "^\[L3D" | tail -3; done; done
