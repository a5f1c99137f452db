// l3d_host.h -- host-side state of libl3dpp_hip.so (the part of class L3DPP::Line3D / L3DPP::View
// the hot path needs), plus small RAII helpers for HIP memory.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/l3dpp_hip.h"
#include "l3d_dev.h"
#include "l3d_kernels.h"

namespace l3d {

void set_error(const std::string& s);
#define L3D_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            l3d::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
            return L3D_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

// Blocks released by a context (l3d_destroy, a buffer that grows) go to a process-wide cache and are handed to the next
// reservation of similar size on the same device instead of back to the runtime: hipFree costs ~0.2 ms and
// hipHostMalloc up to milliseconds on MI355X (tools/alloc_bench.hip), so a process that serves one Line3D object per
// scene would otherwise pay ~10 ms per scene for memory it had a moment ago.  Bounded (blocks; bytes: a quarter of the
// device's memory at most, 1 GiB pinned); l3d_api.hip.
//   * A block may only change hands when nothing is in flight on it.  hipFree / hipHostFree used to see to that (they
//     synchronise the device); block_cache_give does the same unless the caller says the work that touched the block
//     has been waited for (ReleaseSynced: l3d_destroy after it has drained its streams).  Growth is not part of the
//     steady state -- a buffer grows on the first call of a scene larger than any before -- so the wait costs what
//     the hipFree it replaces cost, minus the free.
//   * An allocation that fails empties the cache of its kind (block_cache_trim) and is tried once more: cached blocks
//     are only reused for requests of similar size, so a process that serves scenes of varying size could otherwise
//     run out of memory the runtime still had before the cache existed.  l3d_trim_cache() does the same on demand.
void* block_cache_take(bool pinned, size_t bytes, size_t* got_bytes);   // nullptr: nothing suitable cached
bool block_cache_give(bool pinned, void* p, size_t bytes);              // false: cache full, caller frees
size_t block_cache_trim(int kind);                                      // 0 device, 1 pinned, -1 both; returns bytes freed
struct ReleaseSynced {   // RAII: releases on this thread inside the scope need no device synchronisation
    ReleaseSynced(); ~ReleaseSynced();
};

// device buffer that grows but never shrinks (its block returns to the cache with the context)
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;          // elements
    size_t bytes_ = 0;       // size of the block behind p (a cached block may be larger than asked for)
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        release();
        size_t got = 0;
        if (void* q = block_cache_take(false, n * sizeof(T), &got)) { p = (T*)q; bytes_ = got; cap = got / sizeof(T); return hipSuccess; }
        hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
        if (e != hipSuccess && block_cache_trim(0)) { (void)hipGetLastError(); e = hipMalloc((void**)&p, n * sizeof(T)); }
        if (e == hipSuccess) { cap = n; bytes_ = n * sizeof(T); } else p = nullptr;
        return e;
    }
    // work space that its kernels leave all-zero between launches (k_scan.hip): zeroed when it is (re)allocated
    hipError_t reserve_zeroed(size_t n, hipStream_t st) {
        if (n <= cap) return hipSuccess;
        hipError_t e = reserve(n);
        if (e == hipSuccess) e = hipMemsetAsync(p, 0, cap * sizeof(T), st);
        return e;
    }
    void release() {
        if (p && !block_cache_give(false, p, bytes_)) (void)hipFree(p);
        p = nullptr; cap = 0; bytes_ = 0;
    }
};

// pinned host staging: async H2D copies from it do not stall the stream (pageable sources are copied synchronously)
template <class T>
struct PinnedBuf {
    T* p = nullptr;
    size_t cap = 0, bytes_ = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        release();
        size_t got = 0;
        if (void* q = block_cache_take(true, n * sizeof(T), &got)) { p = (T*)q; bytes_ = got; cap = got / sizeof(T); return hipSuccess; }
        hipError_t e = hipHostMalloc((void**)&p, n * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess && block_cache_trim(1)) { (void)hipGetLastError(); e = hipHostMalloc((void**)&p, n * sizeof(T), hipHostMallocDefault); }
        if (e == hipSuccess) { cap = n; bytes_ = n * sizeof(T); } else p = nullptr;
        return e;
    }
    void release() {
        if (p && !block_cache_give(true, p, bytes_)) (void)hipHostFree(p);
        p = nullptr; cap = 0; bytes_ = 0;
    }
};

// Host -> device through a pinned staging buffer, left out when the device array already holds these very bytes: the
// view table, the pair list and the small tables of phase B are the same from one matchImages call to the next as
// long as the scene is, and each copy is a launch of its own on the stream.  The staging buffer doubles as the record
// of what was sent; `dev_seen` remembers which device allocation received it.  Larger tables are simply sent.
struct UploadTag { const void* dev_seen = nullptr; size_t bytes = 0; };
template <class T, class S>
hipError_t upload_table(DevBuf<T>& dev, PinnedBuf<S>& stage, const void* src, size_t bytes, UploadTag& tag,
                        hipStream_t st, bool* sent = nullptr) {
    if (sent) *sent = false;
    if (!bytes) return hipSuccess;
    const size_t n_stage = (bytes + sizeof(S) - 1) / sizeof(S);
    const bool stage_kept = n_stage <= stage.cap;
    hipError_t e = stage.reserve(n_stage);
    if (e != hipSuccess) return e;
    if (stage_kept && tag.dev_seen == (const void*)dev.p && tag.bytes == bytes && bytes <= (256u << 10) &&
        std::memcmp(stage.p, src, bytes) == 0)
        return hipSuccess;
    std::memcpy(stage.p, src, bytes);
    e = hipMemcpyAsync(dev.p, stage.p, bytes, hipMemcpyHostToDevice, st);
    tag.dev_seen = dev.p; tag.bytes = bytes;
    if (sent) *sent = true;
    return e;
}

// ---- host double 3x3 algebra (mirrors Eigen's fixed-size behaviour; see l3d_dev.h) -------------
struct M3 { double m[9]; };
inline M3 m3_mul(const M3& A, const M3& B) {
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C.m[3 * i + j] = (A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j]) + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
inline M3 m3_t(const M3& A) {
    M3 T;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T.m[3 * i + j] = A.m[3 * j + i];
    return T;
}
inline M3 m3_inv(const M3& A) {  // cofactor formula (Eigen compute_inverse_size3)
    const double* a = A.m;
    const double c00 = a[4] * a[8] - a[5] * a[7], c10 = a[5] * a[6] - a[3] * a[8], c20 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c00 + (a[1] * c10 + a[2] * c20);
    const double id = 1.0 / det;
    return M3{{c00 * id, (a[2] * a[7] - a[1] * a[8]) * id, (a[1] * a[5] - a[2] * a[4]) * id,
               c10 * id, (a[0] * a[8] - a[2] * a[6]) * id, (a[2] * a[3] - a[0] * a[5]) * id,
               c20 * id, (a[1] * a[6] - a[0] * a[7]) * id, (a[0] * a[4] - a[1] * a[3]) * id}};
}

// L3DPP::View (view.h:49-223), hot-path members only
// a set of camera ids as a sorted vector: the neighbour sets are walked and searched for every pair of the list, and a
// std::set is one heap node per element (the first matchImages of a scene of 1 024 views spent 2 ms chasing them)
struct FlatSet {
    std::vector<uint32_t> v;
    bool insert(uint32_t x) {
        auto it = std::lower_bound(v.begin(), v.end(), x);
        if (it != v.end() && *it == x) return false;
        v.insert(it, x);
        return true;
    }
    size_t count(uint32_t x) const { return std::binary_search(v.begin(), v.end(), x) ? 1 : 0; }
    size_t size() const { return v.size(); }
    bool empty() const { return v.empty(); }
    void clear() { v.clear(); }
    std::vector<uint32_t>::const_iterator begin() const { return v.begin(); }
    std::vector<uint32_t>::const_iterator end() const { return v.end(); }
    FlatSet& operator=(const std::set<uint32_t>& s) { v.assign(s.begin(), s.end()); return *this; }
};

struct HostView {
    uint32_t cam = 0, M = 0, index = 0;
    std::vector<float> segs;
    M3 K, R, Kinv, Rt, RtKinv;
    d3 t{0, 0, 0}, C{0, 0, 0}, pp{0, 0, 0};
    uint32_t width = 0, height = 0;
    float initial_median_depth = 0, k = 0, median_depth = 0;
    double coord_max = 0.0;             // largest |coordinate| of the view's segments (inf if any is not finite): kPairFastMath
    std::vector<uint32_t> fixed_nbrs;   // fixed_visual_neighbors_[cam]
    bool by_worldpoints = false;        // added with a worldpoint list (neighbors_by_worldpoints_): neighbours are found
    std::vector<uint32_t> worldpoints;  // from the worldpoint overlap at every matchImages (views2worldpoints_[cam])
    FlatSet visual_nbrs;                // visual_neighbors_[cam]
    // device
    DevBuf<float4> d_seg4;
    DevBuf<SegF> d_segf;
    // pairs touching this view (indices into Ctx::pairs)
    std::vector<uint32_t> out_pairs;    // this view is src, ascending tgt
    std::vector<uint32_t> in_pairs;     // this view is tgt and src < this (inverse matches), ascending src
};

}  // namespace l3d
