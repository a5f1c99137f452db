// Calibration of the VALU issue roof on gfx950 (MI355X): streams of ONE instruction each, 16 independent accumulators,
// measured three ways -- shader cycles per wave (s_memtime), wall time (HIP events), and, when run under
// `rocprofv3 --pmc` (tools/valu_calib.sh), what SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES and the
// per-class instruction counters (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F{32,64}, CVT, INT32, INT64) report for a stream
// whose content is known.  bench.py's `roofline.peak` for k_match_pairs is the issue ceiling of the kernel's own
// instruction mix priced with the cycles measured here (profiles/rNN_valu_calibration.json).
//
// Every kernel: blocks of 64 threads (one wave), grid = 1024 * W blocks (W waves per SIMD when the dispatcher spreads
// them evenly, W = 1, 2, 4, 8), ITERS iterations of 64 instructions.  Printed per (op, W): cycles per instruction and
// SIMD = mean wave time in shader cycles / (instructions per wave * W), the same from wall time with the clock taken
// from s_memtime / s_memrealtime, and wave-instructions per second chip-wide.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../line3dpp_amd/csrc/l3d_dev.h"

#define REP16(M) M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7) M(a8) M(a9) M(a10) M(a11) M(a12) M(a13) M(a14) M(a15)

struct WaveTime { unsigned long long cycles, realtime; };

#define DEF_KERNEL(NAME, TYPE, ...)                                                                              \
    __global__ __launch_bounds__(64) void k_##NAME(WaveTime* out, int iters, TYPE b, TYPE c, TYPE* sink) {       \
        TYPE a0 = (TYPE)(threadIdx.x + 1), a1 = a0 + (TYPE)1, a2 = a0 + (TYPE)2, a3 = a0 + (TYPE)3,               \
             a4 = a0 + (TYPE)4, a5 = a0 + (TYPE)5, a6 = a0 + (TYPE)6, a7 = a0 + (TYPE)7, a8 = a0 + (TYPE)8,       \
             a9 = a0 + (TYPE)9, a10 = a0 + (TYPE)10, a11 = a0 + (TYPE)11, a12 = a0 + (TYPE)12,                    \
             a13 = a0 + (TYPE)13, a14 = a0 + (TYPE)14, a15 = a0 + (TYPE)15;                                       \
        const unsigned long long t0 = clock64(), r0 = wall_clock64();                                             \
        for (int i = 0; i < iters; ++i) {                                                                         \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) { REP16(__VA_ARGS__) }                                  \
        }                                                                                                         \
        const unsigned long long t1 = clock64(), r1 = wall_clock64();                                             \
        TYPE s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;             \
        if (s == (TYPE)123457) *sink = s;                                                                         \
        if (threadIdx.x == 0) out[blockIdx.x] = WaveTime{t1 - t0, r1 - r0};                                       \
    }

// ---- fp32 ----
#define OP_fma_f32(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_mul_f32(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_add_f32(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_max_f32(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_rcp_f32(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define OP_rsq_f32(x) asm volatile("v_rsq_f32 %0, %0" : "+v"(x));
#define OP_sqrt_f32(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
#define OP_exp_f32(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
#define OP_cmp_f32(x) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define OP_cndmask_b32(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");
#define OP_mov_b32(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(b));
// the same without VCC (no compiler-inserted s_nop between the instructions): SGPR-pair results / masks
#define OP_cmp_f32_s(x) { unsigned long long m_; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m_) : "v"(x), "v"(b)); }
#define OP_cndmask_s(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "s"(0x5555555555555555ull));
#define OP_min_f32(x) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_med3_f32(x) asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(x) : "v"(b));
#define OP_add_abs_f32(x) asm volatile("v_add_f32_e64 %0, |%0|, |%1|" : "+v"(x) : "v"(b));
#define OP_readlane(x) { unsigned r_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(r_) : "v"(x)); }
#define OP_writelane(x) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(x) : "s"(7u));
DEF_KERNEL(cmp_f32_s, float, OP_cmp_f32_s)
DEF_KERNEL(cndmask_s, float, OP_cndmask_s)
DEF_KERNEL(min_f32, float, OP_min_f32)
DEF_KERNEL(med3_f32, float, OP_med3_f32)
DEF_KERNEL(add_abs_f32, float, OP_add_abs_f32)
DEF_KERNEL(readlane, float, OP_readlane)
DEF_KERNEL(writelane, float, OP_writelane)
DEF_KERNEL(fma_f32, float, OP_fma_f32)
DEF_KERNEL(mul_f32, float, OP_mul_f32)
DEF_KERNEL(add_f32, float, OP_add_f32)
DEF_KERNEL(max_f32, float, OP_max_f32)
DEF_KERNEL(rcp_f32, float, OP_rcp_f32)
DEF_KERNEL(rsq_f32, float, OP_rsq_f32)
DEF_KERNEL(sqrt_f32, float, OP_sqrt_f32)
DEF_KERNEL(exp_f32, float, OP_exp_f32)
DEF_KERNEL(cmp_f32, float, OP_cmp_f32)
DEF_KERNEL(cndmask_b32, float, OP_cndmask_b32)
DEF_KERNEL(mov_b32, float, OP_mov_b32)
// ---- packed fp32 (two lanes of work per instruction) and fp64: 64-bit registers ----
#define OP_pk_fma_f32(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_pk_mul_f32(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_pk_add_f32(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_fma_f64(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_mul_f64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_add_f64(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_max_f64(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_rcp_f64(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define OP_rsq_f64(x) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
#define OP_sqrt_f64(x) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));
#define OP_cmp_f64(x) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define OP_div_scale_f64(x) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c) : "vcc");
#define OP_div_fmas_f64(x) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c) : "vcc");
#define OP_div_fixup_f64(x) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define OP_lshl_b64(x) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(x));
#define OP_mov_b64(x) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(b));
#define OP_ldexp_f64(x) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x));
#define OP_cmp_f64_s(x) { unsigned long long m_; asm volatile("v_cmp_gt_f64_e64 %0, %1, %2" : "=s"(m_) : "v"(x), "v"(b)); }
#define OP_cmp_class_f64(x) { unsigned long long m_; asm volatile("v_cmp_class_f64_e64 %0, %1, %2" : "=s"(m_) : "v"(x), "v"(0x260u)); }
#define OP_fmac_f64(x) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
DEF_KERNEL(mov_b64, double, OP_mov_b64)
DEF_KERNEL(ldexp_f64, double, OP_ldexp_f64)
DEF_KERNEL(cmp_f64_s, double, OP_cmp_f64_s)
DEF_KERNEL(cmp_class_f64, double, OP_cmp_class_f64)
DEF_KERNEL(fmac_f64, double, OP_fmac_f64)
DEF_KERNEL(pk_fma_f32, double, OP_pk_fma_f32)
DEF_KERNEL(pk_mul_f32, double, OP_pk_mul_f32)
DEF_KERNEL(pk_add_f32, double, OP_pk_add_f32)
DEF_KERNEL(fma_f64, double, OP_fma_f64)
DEF_KERNEL(mul_f64, double, OP_mul_f64)
DEF_KERNEL(add_f64, double, OP_add_f64)
DEF_KERNEL(max_f64, double, OP_max_f64)
DEF_KERNEL(rcp_f64, double, OP_rcp_f64)
DEF_KERNEL(rsq_f64, double, OP_rsq_f64)
DEF_KERNEL(sqrt_f64, double, OP_sqrt_f64)
DEF_KERNEL(cmp_f64, double, OP_cmp_f64)
DEF_KERNEL(div_scale_f64, double, OP_div_scale_f64)
DEF_KERNEL(div_fmas_f64, double, OP_div_fmas_f64)
DEF_KERNEL(div_fixup_f64, double, OP_div_fixup_f64)
DEF_KERNEL(lshl_b64, double, OP_lshl_b64)
// ---- integer ----
#define OP_add_u32(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_and_b32(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_lshl_b32(x) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
#define OP_mul_lo_u32(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define OP_mbcnt(x) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x) : "v"(b));
#define OP_cmp_u32(x) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define OP_lshl_add_u32(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(b));
#define OP_cmp_u32_s(x) { unsigned long long m_; asm volatile("v_cmp_gt_u32_e64 %0, %1, %2" : "=s"(m_) : "v"(x), "v"(b)); }
DEF_KERNEL(lshl_add_u32, unsigned, OP_lshl_add_u32)
DEF_KERNEL(cmp_u32_s, unsigned, OP_cmp_u32_s)
DEF_KERNEL(add_u32, unsigned, OP_add_u32)
DEF_KERNEL(and_b32, unsigned, OP_and_b32)
DEF_KERNEL(lshl_b32, unsigned, OP_lshl_b32)
DEF_KERNEL(mul_lo_u32, unsigned, OP_mul_lo_u32)
DEF_KERNEL(mbcnt, unsigned, OP_mbcnt)
DEF_KERNEL(cmp_u32, unsigned, OP_cmp_u32)
// ---- scalar ALU (one scalar unit per CU, shared by its four SIMDs): the walk of k_match_pairs issues ~0.7 SALU
// instructions per VALU instruction ----
#define OP_s_add_u32(x) asm volatile("s_add_u32 %0, %0, 3" : "+s"(x) : : "scc");
#define OP_s_and_b32(x) asm volatile("s_and_b32 %0, %0, 0x7fffffff" : "+s"(x) : : "scc");
#define OP_s_lshl_b32(x) asm volatile("s_lshl_b32 %0, %0, 1" : "+s"(x) : : "scc");
#define DEF_SKERNEL(NAME, OPM)                                                                                   \
    __global__ __launch_bounds__(64) void k_##NAME(WaveTime* out, int iters, unsigned b, unsigned c, unsigned* sink) { \
        unsigned a0 = b + 1, a1 = b + 2, a2 = b + 3, a3 = b + 4, a4 = b + 5, a5 = b + 6, a6 = b + 7, a7 = b + 8, a8 = b + 9, \
                 a9 = b + 10, a10 = b + 11, a11 = b + 12, a12 = b + 13, a13 = b + 14, a14 = b + 15, a15 = b + 16;  \
        const unsigned long long t0 = clock64(), r0 = wall_clock64();                                             \
        for (int i = 0; i < iters; ++i) {                                                                         \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) { REP16(OPM) }                                          \
        }                                                                                                         \
        const unsigned long long t1 = clock64(), r1 = wall_clock64();                                             \
        unsigned s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;         \
        if (s == 123457u) *sink = s;                                                                              \
        if (threadIdx.x == 0) out[blockIdx.x] = WaveTime{t1 - t0, r1 - r0};                                       \
    }
DEF_SKERNEL(s_add_u32, OP_s_add_u32)
DEF_SKERNEL(s_and_b32, OP_s_and_b32)
DEF_SKERNEL(s_lshl_b32, OP_s_lshl_b32)
// ---- conversions (separate source and destination widths) ----
__global__ __launch_bounds__(64) void k_cvt_f64_f32(WaveTime* out, int iters, float b, float c, float* sink) {
    double a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15;
    float s0 = (float)threadIdx.x + b;
#define OP_cvt(x) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(s0));
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { REP16(OP_cvt) }
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;
    if (s == 123457.0) *sink = (float)s;
    if (threadIdx.x == 0) out[blockIdx.x] = WaveTime{t1 - t0, r1 - r0};
}
__global__ __launch_bounds__(64) void k_cvt_f32_f64(WaveTime* out, int iters, float b, float c, float* sink) {
    float a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15;
    double s0 = (double)threadIdx.x + (double)b;
#define OP_cvt2(x) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x) : "v"(s0));
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { REP16(OP_cvt2) }
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;
    if (s == 123457.0f) *sink = s;
    if (threadIdx.x == 0) out[blockIdx.x] = WaveTime{t1 - t0, r1 - r0};
}
// ---- composite: the compiler's IEEE double division / square root, and the library's own exact pair test ----
__global__ __launch_bounds__(64) void k_c_div_f64(WaveTime* out, int iters, double b, double c, double* sink) {
    double a[16];
    for (int k = 0; k < 16; ++k) a[k] = (double)(threadIdx.x + 1 + k);
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = b / a[k] + c;      // 16 divisions (+ 16 additions) per iteration
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    double s = 0; for (int k = 0; k < 16; ++k) s += a[k];
    if (s == 123457.0) *sink = s;
    if (threadIdx.x == 0) out[blockIdx.x] = WaveTime{t1 - t0, r1 - r0};
}
__global__ __launch_bounds__(64) void k_c_sqrt_f64(WaveTime* out, int iters, double b, double c, double* sink) {
    double a[16];
    for (int k = 0; k < 16; ++k) a[k] = (double)(threadIdx.x + 1 + k);
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = sqrt(a[k]) + c;     // 16 square roots (+ 16 additions) per iteration
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    double s = 0; for (int k = 0; k < 16; ++k) s += a[k];
    if (s == 123457.0) *sink = s;
    if (threadIdx.x == 0) out[blockIdx.x] = WaveTime{t1 - t0, r1 - r0};
}
// one exact_overlap + exact_depths (l3d_dev.h) per lane and iteration on varying inputs: the content of a `drain` of
// k_match_pairs without its LDS work
__global__ __launch_bounds__(64) void k_exact_pair(WaveTime* out, int iters, double b, double c, double* sink) {
    double F[9] = {1e-8, 2e-6 * b, -3e-3, -2.1e-6, 1.5e-8, 4e-3 * c, 2.5e-3, -4.2e-3, 1.0};
    l3d::SegX sx, tx;
    for (int k = 0; k < 3; ++k) {
        sx.r1[k] = 0.5 + 0.01 * k + 1e-4 * threadIdx.x; sx.r2[k] = 0.52 + 0.011 * k; sx.n[k] = 0.3 * (k + 1); sx.rm[k] = 0.51;
        tx.r1[k] = 0.4 + 0.02 * k; tx.r2[k] = 0.45 + 0.013 * k + 1e-4 * threadIdx.x; tx.n[k] = 0.2 * (k + 1); tx.rm[k] = 0.42;
    }
    sx.cn = 1.5; tx.cn = -0.5;
    const double Cs[3] = {0.1, 0.2, 0.3}, Ct[3] = {1.0, -0.5, 0.25};
    float acc = 0.0f;
    float x = 100.0f + (float)threadIdx.x;
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        const float ov = l3d::exact_overlap(F, x, 200.0f, x + 50.0f, 260.0f, 300.0f + acc, 150.0f, 340.0f, 210.0f + x);
        l3d::PairResult res{};
        sx.cn += 1e-9 * ov;
        const bool ok = l3d::exact_depths(sx, tx, Cs, Ct, res);
        acc += ov + (ok ? res.dp1 : 0.0f);
        x += 0.25f;
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    if (acc == 123457.0f) *sink = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = WaveTime{t1 - t0, r1 - r0};
}

struct Op {
    const char* name; const char* cls;
    int insts_per_iter;    // VALU instructions of the measured kind per loop iteration (0: composite, per-iteration cost reported)
    int kind;              // 0 float, 1 double, 2 unsigned, 3 cvt (float args), 4 composite (double args)
    const void* fn;
};

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !std::strcmp(argv[1], "--pmc");   // under rocprofv3: one launch per op (W = 4)
    const int iters = quick ? 2000 : 4000;
    std::vector<Op> ops = {
#define O(n, cls, kind) Op{#n, cls, 64, kind, (const void*)k_##n}
        O(fma_f32, "FMA_F32", 0), O(mul_f32, "MUL_F32", 0), O(add_f32, "ADD_F32", 0), O(max_f32, "other", 0),
        O(rcp_f32, "TRANS_F32", 0), O(rsq_f32, "TRANS_F32", 0), O(sqrt_f32, "TRANS_F32", 0), O(exp_f32, "TRANS_F32", 0),
        O(cmp_f32, "other", 0), O(cndmask_b32, "other", 0), O(mov_b32, "other", 0),
        O(cmp_f32_s, "other", 0), O(cndmask_s, "other", 0), O(min_f32, "other", 0), O(med3_f32, "other", 0), O(add_abs_f32, "ADD_F32 (VOP3, |x| modifiers)", 0), O(readlane, "other", 0), O(writelane, "other", 0),
        O(mov_b64, "other", 1), O(ldexp_f64, "other", 1), O(cmp_f64_s, "other", 1), O(cmp_class_f64, "other", 1), O(fmac_f64, "FMA_F64", 1),
        O(lshl_add_u32, "INT32", 2), O(cmp_u32_s, "INT32", 2),
        O(pk_fma_f32, "FMA_F32 (packed)", 1), O(pk_mul_f32, "MUL_F32 (packed)", 1), O(pk_add_f32, "ADD_F32 (packed)", 1),
        O(fma_f64, "FMA_F64", 1), O(mul_f64, "MUL_F64", 1), O(add_f64, "ADD_F64", 1), O(max_f64, "other", 1),
        O(rcp_f64, "TRANS_F64", 1), O(rsq_f64, "TRANS_F64", 1), O(sqrt_f64, "TRANS_F64", 1), O(cmp_f64, "other", 1),
        O(div_scale_f64, "other", 1), O(div_fmas_f64, "other", 1), O(div_fixup_f64, "other", 1), O(lshl_b64, "INT64", 1),
        O(add_u32, "INT32", 2), O(and_b32, "INT32", 2), O(lshl_b32, "INT32", 2), O(mul_lo_u32, "INT32", 2),
        O(mbcnt, "INT32", 2), O(cmp_u32, "INT32", 2),
        O(cvt_f64_f32, "CVT", 3), O(cvt_f32_f64, "CVT", 3),
        O(s_add_u32, "SALU", 2), O(s_and_b32, "SALU", 2), O(s_lshl_b32, "SALU", 2),
#undef O
        Op{"c_div_f64", "composite: 16 x (IEEE double division + add) per iteration", 0, 4, (const void*)k_c_div_f64},
        Op{"c_sqrt_f64", "composite: 16 x (IEEE double sqrt + add) per iteration", 0, 4, (const void*)k_c_sqrt_f64},
        Op{"exact_pair", "composite: exact_overlap + exact_depths (l3d_dev.h) per iteration", 0, 4, (const void*)k_exact_pair},
    };
    WaveTime* d_out = nullptr; double* d_sink = nullptr;
    const int max_blocks = 1024 * 8;
    CHECK(hipMalloc(&d_out, sizeof(WaveTime) * max_blocks));
    CHECK(hipMalloc(&d_sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<WaveTime> h(max_blocks);
    std::printf("{\"iters\": %d, \"ops\": [\n", iters);
    bool first = true;
    for (const Op& op : ops) {
        for (int W : {1, 2, 4, 8}) {
            if (quick && W != 4) continue;
            const int blocks = 1024 * W;
            const int it = op.insts_per_iter ? iters : iters / 8;
            float fb = 1.0000001f, fc = 1e-9f; double db = 1.0000001, dc = 1e-9; unsigned ub = 3u, uc = 5u;
            void* args_f[] = {&d_out, (void*)&it, &fb, &fc, &d_sink};
            void* args_d[] = {&d_out, (void*)&it, &db, &dc, &d_sink};
            void* args_u[] = {&d_out, (void*)&it, &ub, &uc, &d_sink};
            void** args = op.kind == 0 || op.kind == 3 ? args_f : op.kind == 2 ? args_u : args_d;
            for (int rep = 0; rep < (quick ? 1 : 2); ++rep) {   // second launch is the measured one (warm clocks)
                CHECK(hipEventRecord(e0, 0));
                CHECK(hipLaunchKernel(op.fn, dim3(blocks), dim3(64), args, 0, 0));
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
            }
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_out, sizeof(WaveTime) * blocks, hipMemcpyDeviceToHost));
            double cyc = 0, rt = 0, cmax = 0;
            for (int k = 0; k < blocks; ++k) { cyc += (double)h[k].cycles; rt += (double)h[k].realtime; cmax = std::max(cmax, (double)h[k].cycles); }
            cyc /= blocks; rt /= blocks;
            const double clk_ghz = rt > 0 ? cyc / rt * 0.1 : 0.0;                 // s_memrealtime ticks at 100 MHz
            const double n_inst = op.insts_per_iter ? (double)it * op.insts_per_iter : (double)it;
            const double cyc_per_inst = cyc / (n_inst * W);                        // per SIMD, W waves sharing it
            const double cyc_wall = ms * 1e-3 * clk_ghz * 1e9 * 1024.0 / (n_inst * blocks);
            std::printf("%s {\"op\": \"%s\", \"class\": \"%s\", \"waves_per_simd\": %d, \"unit\": \"%s\", \"cycles_per_unit_simd\": %.3f, "
                        "\"cycles_per_unit_simd_wall\": %.3f, \"kernel_ms\": %.4f, \"clock_ghz\": %.3f, \"G_units_per_s\": %.2f, "
                        "\"max_over_mean_wave_cycles\": %.3f}",
                        first ? " " : ",\n ", op.name, op.cls, W, op.insts_per_iter ? "wave64 instruction" : "iteration",
                        cyc_per_inst, cyc_wall, ms, clk_ghz, n_inst * blocks / (ms * 1e-3) / 1e9, cmax / std::max(cyc, 1.0));
            first = false;
        }
    }
    std::printf("\n]}\n");
    return 0;
}
