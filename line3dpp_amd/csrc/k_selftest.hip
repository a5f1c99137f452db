// k_selftest.hip -- device self-test of the arithmetic shortcuts of l3d_dev.h (test hook behind l3d_selftest_arith):
// the unscaled IEEE division / square root (rcp_refined + div_by, sqrt_unscaled) against the compiler's own expansions
// of a / b and sqrt(x) on the device, bit for bit, over operands drawn across the range the host admits for
// kPairFastMath (and beyond the reference's guards: denominators down to L3D_EPS, exact zeros in the numerator).
#include "l3d_ctx.h"

namespace l3d {
namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// a double with uniformly random mantissa and an exponent of 10 drawn from [lo10, hi10], random sign
__device__ __forceinline__ double draw(uint64_t& st, int lo10, int hi10) {
    st = mix64(st);
    const double m = 1.0 + (double)(st >> 12) * (1.0 / 4503599627370496.0);   // [1, 2)
    st = mix64(st);
    const int e10 = lo10 + (int)(st % (uint64_t)(hi10 - lo10 + 1));
    const double v = m * pow(10.0, (double)e10);
    return (st >> 40) & 1 ? -v : v;
}

__global__ void k_selftest_arith(uint64_t n, uint64_t seed, unsigned long long* out) {
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad_div = 0, bad_sqrt = 0, bad_div2 = 0;
    for (uint64_t i = i0; i < n; i += stride) {
        uint64_t st = seed ^ (i * 0xD1342543DE82EF95ull);
        const int kind = (int)(i % 4);
        // numerators from exact zero / tiny up to 1e60, denominators from 1e-12 (the reference's guards) to 1e52
        double a = kind == 3 ? 0.0 : draw(st, kind == 0 ? -3 : -72, kind == 0 ? 8 : 60);
        double a2 = draw(st, -20, 20);
        const double b = draw(st, kind == 0 ? -3 : -12, kind == 0 ? 8 : 52);
        if (kind == 2) a = b * (1.0 + (double)((int)(st % 9) - 4) * 2.220446049250313e-16);   // quotients next to 1
        const double q_ref = a / b, q2_ref = a2 / b;
        double q1, q2;
        L3D_DIV2(true, a, a2, b, q1, q2);
        const double qs = L3D_DIV(true, a, b);
        bad_div2 += (__double_as_longlong(q1) != __double_as_longlong(q_ref)) + (__double_as_longlong(q2) != __double_as_longlong(q2_ref));
        bad_div += __double_as_longlong(qs) != __double_as_longlong(q_ref);
        const double x = kind == 3 ? 0.0 : fabs(draw(st, kind == 0 ? -2 : -180, kind == 0 ? 14 : 140));
        bad_sqrt += __double_as_longlong(L3D_SQRT(true, x)) != __double_as_longlong(sqrt(x));
        // perfect squares and their neighbours (ties of the final rounding)
        const double r = floor(fabs(draw(st, 0, 7))), sq = r * r;
        bad_sqrt += __double_as_longlong(L3D_SQRT(true, sq)) != __double_as_longlong(sqrt(sq));
    }
    atomicAdd(&out[0], bad_div); atomicAdd(&out[1], bad_div2); atomicAdd(&out[2], bad_sqrt);
}

}  // namespace
}  // namespace l3d

// counts[3]: single divisions, paired divisions (shared reciprocal), square roots whose bits differ from the compiler's
extern "C" int l3d_selftest_arith(int device, uint64_t n, uint64_t seed, uint64_t counts[3]) {
    if (!counts) return fail(L3D_ERR_ARG, "null argument");
    if (hipSetDevice(device) != hipSuccess) return fail(L3D_ERR_HIP, "hipSetDevice failed: no usable HIP device");
    unsigned long long* d = nullptr;
    L3D_HIP_CHECK(hipMalloc((void**)&d, 24));
    L3D_HIP_CHECK(hipMemset(d, 0, 24));
    hipLaunchKernelGGL(l3d::k_selftest_arith, dim3(2048), dim3(256), 0, 0, n, seed, d);
    L3D_HIP_CHECK(hipGetLastError());
    unsigned long long h[3];
    L3D_HIP_CHECK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    counts[0] = h[0]; counts[1] = h[1]; counts[2] = h[2];
    return L3D_OK;
}
