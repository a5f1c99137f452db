"""Which asynchronous copy of a fresh process pays the runtime's lazy set-up?  Times every hipMemcpyAsync (+ its
synchronisation) of a sequence of pinned <-> device copies issued through the runtime directly (ctypes)."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "lib":
    from line3dpp_amd.api import Line3D
    g = Line3D()
hip = C.CDLL("libamdhip64.so")
d = C.c_void_p(); h = C.c_void_p()
N = 8 << 20
assert hip.hipMalloc(C.byref(d), C.c_size_t(N)) == 0 and hip.hipHostMalloc(C.byref(h), C.c_size_t(N), 0) == 0
hip.hipDeviceSynchronize()
seq = []
for rep in range(40):
    for kind, s in ((1, 3952), (1, 16640), (1, 17680), (1, 10400), (2, 136), (2, 16768)):
        seq.append((kind, s))
slow = []
t_all = time.time()
for i, (kind, s) in enumerate(seq):
    t0 = time.perf_counter()
    a, b = (d, h) if kind == 1 else (h, d)
    assert hip.hipMemcpyAsync(a, b, C.c_size_t(s), kind, None) == 0
    t1 = time.perf_counter()
    if (i % 6) == 5:
        hip.hipStreamSynchronize(None)
    t2 = time.perf_counter()
    if t2 - t0 > 0.5e-3:
        slow.append((i, "H2D" if kind == 1 else "D2H", s, round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2), round((t0 - t_all) * 1e3, 1)))
print(mode, "copies", len(seq), "slow (index, dir, bytes, issue ms, sync ms, at ms):", slow)
