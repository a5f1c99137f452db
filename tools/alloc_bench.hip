// What a fresh context pays for its memory on this box: hipMalloc / hipFree / hipHostMalloc / hipMemsetAsync / event and
// stream creation, timed on the host (microseconds, median of 5).  Behind the cold-call design of l3d_host.h (arena).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double med(F f) { std::vector<double> v; for (int i = 0; i < 5; ++i) v.push_back(f()); std::sort(v.begin(), v.end()); return v[2]; }
int main() {
    (void)hipFree(nullptr);
    std::printf("{");
    for (size_t mb : {1ul, 16ul, 64ul, 256ul, 1024ul, 4096ul}) {
        double tf = 0;
        const double ta = med([&] { void* p; double t0 = now_us(); (void)hipMalloc(&p, mb << 20); double t1 = now_us(); (void)hipFree(p); tf = now_us() - t1; return t1 - t0; });
        std::printf("\"hipMalloc_%zuMiB_us\": %.1f, \"hipFree_%zuMiB_us\": %.1f, ", mb, ta, mb, tf);
    }
    for (size_t kb : {64ul, 1024ul, 16384ul}) {
        const double ta = med([&] { void* p; double t0 = now_us(); (void)hipHostMalloc(&p, kb << 10, hipHostMallocDefault); double t1 = now_us(); (void)hipHostFree(p); return t1 - t0; });
        std::printf("\"hipHostMalloc_%zuKiB_us\": %.1f, ", kb, ta);
    }
    {   // 40 small device buffers, as a context without an arena makes them
        const double t = med([&] { std::vector<void*> ps(40); double t0 = now_us(); for (auto& p : ps) (void)hipMalloc(&p, 4 << 20); double t1 = now_us(); for (auto& p : ps) (void)hipFree(p); return t1 - t0; });
        std::printf("\"hipMalloc_40x4MiB_us\": %.1f, ", t);
    }
    const double te = med([&] { hipEvent_t e; double t0 = now_us(); (void)hipEventCreate(&e); double t1 = now_us(); (void)hipEventDestroy(e); return t1 - t0; });
    const double ts = med([&] { hipStream_t s; double t0 = now_us(); (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); double t1 = now_us(); (void)hipStreamDestroy(s); return t1 - t0; });
    void* p; (void)hipMalloc(&p, 1ul << 30);
    const double tm = med([&] { double t0 = now_us(); (void)hipMemsetAsync(p, 0, 1ul << 30, 0); (void)hipDeviceSynchronize(); return now_us() - t0; });
    std::printf("\"hipEventCreate_us\": %.1f, \"hipStreamCreate_us\": %.1f, \"memset_1GiB_us\": %.1f}\n", te, ts, tm);
    return 0;
}
