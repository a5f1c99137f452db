// k_rdd.hip -- replicator-dynamics diffusion of the affinity matrix.
//
// Replaces Line3D::performRDD (line3D.cc:2026-2076) + replicator_dynamics_diffusion_GPU (cudawrapper.cu:708-766)
// with its kernels K_sparseMat_row_normalization (:432-477) and K_sparseMat_diffusion_step (:480-544).  The
// reference keeps the matrix as float4 (i, j, w, 0) COO lists, once sorted by column (W) and once by row (P, P'),
// finds rows through start_indices and stores results through a linear search.  Here the matrix is ONE CSR
// structure built on the device (64-bit radix sort of (i,j) keys, row pointers and the position of every
// transposed entry by binary search) with three value arrays; results are identical to the reference's
// algorithm entry by entry, including its quirk: the "row of P times column of W" product walks the two entry
// lists in LOCKSTEP (k-th entry of row r of P times k-th entry of column c of W, cudawrapper.cu:500-515) without
// matching indices, and stops at the shorter list.  A drop-in has to reproduce that, not fix it.
//
//   P  <- A (row sorted);  P <- rownorm(P)
//   10 x { P'[j,i] <- max(eps, P[i,j] * sum_k P_row(j)[k] * W_col(i)[k]);  swap(P, P');  rownorm(P) except last }
//   A_[i,j] <- min(P[i,j], P[j,i])   emitted in (i, j) order (the std::map order of performRDD)
//
// Column c of W sorted by row is, entry for entry, { A[r,c] : r in N(c) } = the TRANSPOSED values of row c of the
// CSR (the pattern of A_ is symmetric: computingAffinityMatrix always pushes (i,j) and (j,i)), so no second,
// column-sorted copy exists.  A missing transposed entry (foreign input) behaves as in the reference: nothing is
// stored for it and the symmetrisation keeps w12.
//
// Roofline: per iteration every entry reads min(deg i, deg j) pairs of 4-byte values that neighbouring entries
// of the same row re-read (L2 hits) -- compulsory HBM traffic is 3 value arrays + indices = ~20 B per entry and
// iteration; at the sizes of this path (C1: 2.7e5 entries) the step is launch-latency bound, so the 20 launches
// are kept minimal rather than fused.
#include <hipcub/hipcub.hpp>

#include "l3d_dev.h"
#include "l3d_kernels.h"
#include "../../include/l3dpp_hip.h"

namespace l3d {

namespace {

constexpr float kEpsGpu = 1e-12f;   // L3D_EPS_GPU, cudawrapper.h:50

__global__ void k_rdd_keys(const l3d_cledge* __restrict__ e, uint32_t nnz, uint64_t* __restrict__ keys,
                           uint32_t* __restrict__ idx) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    keys[t] = ((uint64_t)(uint32_t)e[t].i_ << 32) | (uint32_t)e[t].j_;
    idx[t] = t;
}

__device__ __forceinline__ uint32_t lower_bound64(const uint64_t* __restrict__ a, uint32_t lo, uint32_t hi, uint64_t key) {
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (a[m] < key) lo = m + 1; else hi = m; }
    return lo;
}

__global__ void k_rdd_rowptr(const uint64_t* __restrict__ keys, uint32_t nnz, uint32_t n_rows,
                             uint32_t* __restrict__ row_ptr) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    row_ptr[r] = (r == n_rows) ? nnz : lower_bound64(keys, 0, nnz, (uint64_t)r << 32);
}

// per entry (sorted position t): column, initial value, position of the transposed entry
__global__ void k_rdd_csr(const l3d_cledge* __restrict__ e, const uint64_t* __restrict__ keys,
                          const uint32_t* __restrict__ idx, const uint32_t* __restrict__ row_ptr, uint32_t nnz,
                          uint32_t n_rows, uint32_t* __restrict__ row, uint32_t* __restrict__ col,
                          float* __restrict__ a_val, uint32_t* __restrict__ tpos) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    const uint64_t k = keys[t];
    const uint32_t i = (uint32_t)(k >> 32), j = (uint32_t)k;
    row[t] = i; col[t] = j;
    a_val[t] = e[idx[t]].w_;
    uint32_t tp = kEmpty;
    if (j < n_rows) {
        const uint64_t kt = ((uint64_t)j << 32) | i;
        const uint32_t p = lower_bound64(keys, row_ptr[j], row_ptr[j + 1], kt);
        if (p < row_ptr[j + 1] && keys[p] == kt) tp = p;
    }
    tpos[t] = tp;
}

// W sorted by column, read as "transposed values in row order": wt[t] = A[col[t], row[t]]
__global__ void k_rdd_wt(const float* __restrict__ a_val, const uint32_t* __restrict__ tpos, uint32_t nnz,
                         float* __restrict__ wt, float* __restrict__ p, float* __restrict__ p2) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    const float a = a_val[t];
    wt[t] = tpos[t] != kEmpty ? a_val[tpos[t]] : a;
    p[t] = a; p2[t] = a;   // P and P' both start as copies of W (cudawrapper.cu:721-725)
}

// K_sparseMat_row_normalization: sequential float sum in entry order, clamp, divide
__global__ void k_rdd_rownorm(const uint32_t* __restrict__ row_ptr, uint32_t n_rows, float* __restrict__ p) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t b = row_ptr[r], e = row_ptr[r + 1];
    float sum = 0.0f;
    for (uint32_t t = b; t < e; ++t) sum += p[t];
    if (sum < kEpsGpu) sum = kEpsGpu;
    for (uint32_t t = b; t < e; ++t) p[t] /= sum;
}

// K_sparseMat_diffusion_step for entry (i, j) of P: r = j, c = i (transposed); lockstep product of row r of P
// and column c of W; times P[i,j]; clamp; stored at (r, c) of P'
__global__ void k_rdd_step(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ row,
                           const uint32_t* __restrict__ col, const uint32_t* __restrict__ tpos,
                           const float* __restrict__ p, const float* __restrict__ wt, uint32_t nnz, uint32_t n_rows,
                           float* __restrict__ p_out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    const uint32_t c = row[t], r = col[t];
    const uint32_t tp = tpos[t];
    if (tp == kEmpty) return;               // the reference's linear search finds no (r, c): nothing stored
    const uint32_t pb = row_ptr[r], pn = row_ptr[r + 1] - pb;   // row r of P
    const uint32_t wb = row_ptr[c], wn = row_ptr[c + 1] - wb;   // column c of W
    const uint32_t n = pn < wn ? pn : wn;
    float mul = 0.0f;
    for (uint32_t k = 0; k < n; ++k) mul += p[pb + k] * wt[wb + k];
    mul *= p[t];
    if (mul < kEpsGpu) mul = kEpsGpu;
    p_out[tp] = mul;
}

// performRDD's symmetrisation, emitted in (i, j) order
__global__ void k_rdd_sym(const uint32_t* __restrict__ row, const uint32_t* __restrict__ col,
                          const uint32_t* __restrict__ tpos, const float* __restrict__ p, uint32_t nnz,
                          l3d_cledge* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    const float w12 = p[t];
    const float w21 = tpos[t] != kEmpty ? p[tpos[t]] : w12;
    l3d_cledge e;
    e.i_ = (int32_t)row[t]; e.j_ = (int32_t)col[t]; e.w_ = fminf(w12, w21);
    out[t] = e;
}

}  // namespace

size_t rdd_workspace_bytes(uint32_t nnz, uint32_t n_rows) {
    size_t sort_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)nnz);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return al(sort_tmp) + 2 * al((size_t)nnz * 8) + 2 * al((size_t)nnz * 4) /* idx in/out */ +
           al(((size_t)n_rows + 1) * 4) + 3 * al((size_t)nnz * 4) /* row col tpos */ + 5 * al((size_t)nnz * 4) /* a wt p p2 spare */;
}

// edges_in: nnz CLEdges (any order), ids < n_rows.  edges_out: nnz CLEdges, (i, j) ascending.
hipError_t launch_rdd(const l3d_cledge* edges_in, uint32_t nnz, uint32_t n_rows, uint32_t iterations,
                      l3d_cledge* edges_out, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (!nnz || !n_rows) return hipSuccess;
    if (workspace_bytes < rdd_workspace_bytes(nnz, n_rows)) return hipErrorInvalidValue;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t sort_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)nnz);
    char* w = (char*)workspace;
    auto take = [&](size_t b) { char* p = w; w += al(b); return p; };
    void* d_tmp = take(sort_tmp);
    uint64_t* keys_in = (uint64_t*)take((size_t)nnz * 8);
    uint64_t* keys = (uint64_t*)take((size_t)nnz * 8);
    uint32_t* idx_in = (uint32_t*)take((size_t)nnz * 4);
    uint32_t* idx = (uint32_t*)take((size_t)nnz * 4);
    uint32_t* row_ptr = (uint32_t*)take(((size_t)n_rows + 1) * 4);
    uint32_t* row = (uint32_t*)take((size_t)nnz * 4);
    uint32_t* col = (uint32_t*)take((size_t)nnz * 4);
    uint32_t* tpos = (uint32_t*)take((size_t)nnz * 4);
    float* a_val = (float*)take((size_t)nnz * 4);
    float* wt = (float*)take((size_t)nnz * 4);
    float* p = (float*)take((size_t)nnz * 4);
    float* p2 = (float*)take((size_t)nnz * 4);
    const dim3 blk(256), ge((nnz + 255) / 256), gr((n_rows + 256) / 256);
    hipLaunchKernelGGL(k_rdd_keys, ge, blk, 0, st, edges_in, nnz, keys_in, idx_in);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(d_tmp, sort_tmp, keys_in, keys, idx_in, idx, (int)nnz, 0, 64, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_rdd_rowptr, gr, blk, 0, st, keys, nnz, n_rows, row_ptr);
    hipLaunchKernelGGL(k_rdd_csr, ge, blk, 0, st, edges_in, keys, idx, row_ptr, nnz, n_rows, row, col, a_val, tpos);
    hipLaunchKernelGGL(k_rdd_wt, ge, blk, 0, st, a_val, tpos, nnz, wt, p, p2);
    hipLaunchKernelGGL(k_rdd_rownorm, gr, blk, 0, st, row_ptr, n_rows, p);
    for (uint32_t it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(k_rdd_step, ge, blk, 0, st, row_ptr, row, col, tpos, p, wt, nnz, n_rows, p2);
        float* t = p; p = p2; p2 = t;
        if (it + 1 < iterations) hipLaunchKernelGGL(k_rdd_rownorm, gr, blk, 0, st, row_ptr, n_rows, p);
    }
    hipLaunchKernelGGL(k_rdd_sym, ge, blk, 0, st, row, col, tpos, p, nnz, edges_out);
    return hipGetLastError();
}

}  // namespace l3d


// ---- start-up (l3d_create): the runtime loads a translation unit's code object at the first launch of one of its
// kernels (~0.6 ms each, measured on the first matchImages of a process); an empty launch pays that at context creation
namespace l3d {
namespace { __global__ void k_warm_rdd() {} }
hipError_t warm_rdd(hipStream_t st) {
    hipLaunchKernelGGL(k_warm_rdd, dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}
}  // namespace l3d
