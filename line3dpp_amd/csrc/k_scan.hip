// k_scan.hip -- exclusive scans of phase B and of the affinity fill, ONE launch each.
//
// Phase B needs list offsets (a scan over the ~10^5..10^6 2D segments) four times per matchImages; as block sums /
// scan of sums / down-sweep that was twelve launches of a few microseconds of work each.  Here a scan is a single
// pass with decoupled look-back: a tile (1024 threads x 4 items) scans itself, publishes its aggregate at once, adds
// up the aggregates of the tiles before it -- 64 of them per step, one per lane of its first wave, stopping at the
// first one that already knows its inclusive prefix -- and publishes its own inclusive prefix.  A workgroup takes its
// tile index from a ticket counter when it starts, so a tile only ever waits for tiles whose workgroups are already
// running -- whatever order the hardware dispatches the workgroups of a launch in (until round 4 the tile index was
// blockIdx.x, which relied on in-order dispatch).
//
// Work space: [0] tiles finished, [1] tickets handed out, then the state: one 64-bit word per tile and 32-bit lane of the element type (kind << 32 | value; kind 0 = nothing yet,
// 1 = aggregate, 2 = inclusive prefix), written and read whole with agent-scope atomics, so no fence is needed.  The
// last tile to finish zeroes the state again: the work space is all-zero between launches (it is zeroed when it is
// allocated, DevBuf::reserve_zeroed) and no launch needs a memset of its own.
//
// uint64 elements are two packed 32-bit counters (surviving matches | best hypotheses per segment, k_lists.hip: k_seg_filter), scanned
// as two lanes at once; their totals stay below 2^32 (l3d_api.hip checks the slot count), so the halves never carry.
#include "l3d_dev.h"
#include "l3d_kernels.h"

namespace l3d {

namespace {

constexpr uint32_t kScanBlock = 1024, kScanItems = 4, kScanTile = kScanBlock * kScanItems;
constexpr unsigned long long kAgg = 1ull << 32, kIncl = 2ull << 32;

template <class T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* wsum /*[17]*/, T& block_total) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    T x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    if (tid == 0) {
        T acc = 0;
        for (uint32_t w = 0; w < kScanBlock / 64; ++w) { const T t = wsum[w]; wsum[w] = acc; acc += t; }
        wsum[16] = acc;
    }
    __syncthreads();
    block_total = wsum[16];
    return wsum[wave] + x - v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ws: [0] tiles finished, [1] next ticket, [2 + tile * LANES + l] state of lane l of a tile
template <class T>
__global__ __launch_bounds__(kScanBlock) void k_scan(const T* __restrict__ in, uint32_t n, T* __restrict__ out,
                                                     T* __restrict__ total_out, unsigned long long* __restrict__ ws) {
    constexpr uint32_t LANES = sizeof(T) / 4;
    __shared__ T wsum[17];
    __shared__ T s_prefix;
    __shared__ uint32_t s_last, s_tile;
    const uint32_t nb = gridDim.x, tid = threadIdx.x, lane = tid & 63u;
    if (tid == 0) s_tile = (uint32_t)__hip_atomic_fetch_add(&ws[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * kScanTile + tid * kScanItems;
    T a[kScanItems];
    T v = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; ++k) { a[k] = (base + k < n) ? in[base + k] : (T)0; v += a[k]; }
    T agg;
    T ex = block_exclusive_scan(v, wsum, agg);
    unsigned long long* state = ws + 2;
    if (tid < 64) {
        if (lane < LANES)
            __hip_atomic_store(&state[(size_t)tile * LANES + lane],
                               (tile == 0 ? kIncl : kAgg) | (uint32_t)((unsigned long long)agg >> (32 * lane)),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long prefix = 0;
        if (tile > 0) {
#pragma unroll
            for (uint32_t l = 0; l < LANES; ++l) {
                uint32_t acc = 0;
                for (int idx = (int)tile - 1;; idx -= 64) {
                    const int t = idx - (int)lane;                 // tiles before the first one: prefix 0, known
                    unsigned long long s = kIncl;
                    if (t >= 0) {
                        do s = __hip_atomic_load(&state[(size_t)t * LANES + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        while ((s >> 32) == 0);
                    }
                    const uint64_t incl = __builtin_amdgcn_ballot_w64((s >> 32) == 2);
                    if (incl) {
                        const uint32_t first = (uint32_t)__builtin_ctzll(incl);
                        acc += wave_sum(lane <= first ? (uint32_t)s : 0u);
                        break;
                    }
                    acc += wave_sum((uint32_t)s);
                }
                prefix |= (unsigned long long)acc << (32 * l);
            }
            if (lane < LANES)
                __hip_atomic_store(&state[(size_t)tile * LANES + lane],
                                   kIncl | (uint32_t)((prefix + (unsigned long long)agg) >> (32 * lane)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_prefix = (T)prefix;
    }
    __syncthreads();
    ex += s_prefix;
#pragma unroll
    for (uint32_t k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += a[k];
    }
    if (tile == nb - 1 && tid == 0) {
        const T total = s_prefix + agg;
        out[n] = total;
        if (total_out) *total_out = total;
    }
    // the last tile to get here leaves the work space zeroed for the next scan.  The count is a release / acquire pair at
    // agent scope: a tile's state stores are ordered before its increment and the last tile's zeroing after it has seen
    // every increment, so no straggling state store can land behind the zeroing (relaxed operations on different
    // addresses carry no order of their own; one s_waitcnt per tile).
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&ws[0], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nb - 1;
    __syncthreads();
    if (s_last) {
        for (uint32_t i = tid; i < nb * LANES; i += kScanBlock)
            __hip_atomic_store(&state[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(&ws[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ws[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <class T>
hipError_t launch(const T* in, uint32_t n, T* out, unsigned long long* ws, T* total, hipStream_t st) {
    const uint32_t nb = std::max(1u, (n + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL((k_scan<T>), dim3(nb), dim3(kScanBlock), 0, st, in, n, out, total, ws);
    return hipGetLastError();
}

}  // namespace

// 64-bit words of work space a scan of n elements needs (zeroed once: DevBuf::reserve_zeroed)
size_t scan_ws_words(size_t n, uint32_t bytes_per_element) { return 4 + (n / kScanTile + 1) * (bytes_per_element / 4); }

// out[0..n] = exclusive scan of in[0..n) (out[n] = total, also stored to *total).  in and out may be the same array.
hipError_t launch_scan(const uint32_t* in, uint32_t n, uint32_t* out, unsigned long long* ws, uint32_t* total, hipStream_t st) {
    return launch<uint32_t>(in, n, out, ws, total, st);
}
hipError_t launch_scan64(const unsigned long long* in, uint32_t n, unsigned long long* out, unsigned long long* ws,
                         unsigned long long* total, hipStream_t st) {
    return launch<unsigned long long>(in, n, out, ws, total, st);
}

}  // namespace l3d


// ---- start-up (l3d_create): the runtime loads a translation unit's code object at the first launch of one of its
// kernels (~0.6 ms each, measured on the first matchImages of a process); an empty launch pays that at context creation
namespace l3d {
namespace { __global__ void k_warm_scan() {} }
hipError_t warm_scan(hipStream_t st) {
    hipLaunchKernelGGL(k_warm_scan, dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}
}  // namespace l3d
