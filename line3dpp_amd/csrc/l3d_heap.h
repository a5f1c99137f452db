// l3d_heap.h -- the kNN selection order of the reference, reproduced exactly.
//
// Line3D::matchingCPU (line3D.cc:982-1007) pushes every accepted match of a source segment, in ascending target
// order, into std::priority_queue<Match, std::vector<Match>, Match_kNN> (commons.h:217-231; the comparator looks at
// overlap_score_ only) and pops kNN of them.  For distinct overlaps that is "the kNN largest, descending"; for EQUAL
// overlaps both the popped set (a tie at the kNN-th place) and the order are whatever libstdc++'s binary heap
// produces.  These two functions are std::push_heap / std::pop_heap of libstdc++ (bits/stl_heap.h: __push_heap,
// __adjust_heap, __pop_heap) on (overlap, target) pairs kept in two parallel arrays, so that a row with ties can be
// replayed bit for bit (k_match.hip: k_match_tied_rows).  tests/test_heap_order.py pins them against
// std::priority_queue itself on tie-heavy sequences.  Plain C++ (g++ compiles it for that test); the pointer types
// are template parameters so that device code can pass LDS or global pointers.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define L3D_HEAP_HD __host__ __device__ __forceinline__
#else
#define L3D_HEAP_HD inline
#endif

namespace l3d {

// std::push_heap after push_back: n = number of elements BEFORE the push
template <class OvPtr, class IxPtr>
L3D_HEAP_HD void heap_push(OvPtr ov, IxPtr ix, uint32_t n, float v, uint32_t x) {
    uint32_t hole = n;
    while (hole > 0) {
        const uint32_t parent = (hole - 1) / 2;
        if (!(ov[parent] < v)) break;
        ov[hole] = ov[parent]; ix[hole] = ix[parent];
        hole = parent;
    }
    ov[hole] = v; ix[hole] = x;
}

// top() + std::pop_heap + pop_back: n = number of elements BEFORE the pop (> 0)
template <class OvPtr, class IxPtr>
L3D_HEAP_HD void heap_pop(OvPtr ov, IxPtr ix, uint32_t n, float& top_v, uint32_t& top_x) {
    top_v = ov[0]; top_x = ix[0];
    const uint32_t len = n - 1;            // __adjust_heap works on [first, last - 1)
    if (len == 0) return;
    const float v = ov[len]; const uint32_t x = ix[len];
    uint32_t hole = 0, second = 0;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (ov[second] < ov[second - 1]) --second;
        ov[hole] = ov[second]; ix[hole] = ix[second];
        hole = second;
    }
    if ((len & 1u) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        ov[hole] = ov[second - 1]; ix[hole] = ix[second - 1];
        hole = second - 1;
    }
    while (hole > 0) {                     // __push_heap(first, hole, 0, value)
        const uint32_t parent = (hole - 1) / 2;
        if (!(ov[parent] < v)) break;
        ov[hole] = ov[parent]; ix[hole] = ix[parent];
        hole = parent;
    }
    ov[hole] = v; ix[hole] = x;
}

// The same two operations on PACKED entries: overlap bits in the high word, target in the low word, one 64-bit access
// per element instead of two 32-bit ones (the replay kernel's heap lives in LDS and one thread walks it: every access
// is a dependent round trip).  The overlaps that reach the heap are positive floats, whose IEEE bit patterns order
// like the values, so the comparison is on the high words as unsigned integers -- the low word (target) must not
// take part, the reference's comparator does not look at it.
L3D_HEAP_HD uint64_t heap_pack(float ov, uint32_t ix) {
    union { float f; uint32_t u; } c; c.f = ov;
    return (uint64_t)c.u << 32 | ix;
}
L3D_HEAP_HD float heap_overlap(uint64_t e) {
    union { float f; uint32_t u; } c; c.u = (uint32_t)(e >> 32);
    return c.f;
}
L3D_HEAP_HD bool heap_less(uint64_t a, uint64_t b) { return (uint32_t)(a >> 32) < (uint32_t)(b >> 32); }

template <class Ptr>
L3D_HEAP_HD void heap_push_packed(Ptr h, uint32_t n, uint64_t e) {
    uint32_t hole = n;
    while (hole > 0) {
        const uint32_t parent = (hole - 1) / 2;
        const uint64_t p = h[parent];
        if (!heap_less(p, e)) break;
        h[hole] = p;
        hole = parent;
    }
    h[hole] = e;
}

template <class Ptr>
L3D_HEAP_HD uint64_t heap_pop_packed(Ptr h, uint32_t n) {
    const uint64_t top = h[0];
    const uint32_t len = n - 1;
    if (len == 0) return top;
    const uint64_t e = h[len];
    uint32_t hole = 0, second = 0;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        const uint64_t r = h[second], l = h[second - 1];    // independent loads: one round trip per level
        uint64_t pick = r;
        if (heap_less(r, l)) { --second; pick = l; }
        h[hole] = pick;
        hole = second;
    }
    if ((len & 1u) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        h[hole] = h[second - 1];
        hole = second - 1;
    }
    while (hole > 0) {
        const uint32_t parent = (hole - 1) / 2;
        const uint64_t p = h[parent];
        if (!heap_less(p, e)) break;
        h[hole] = p;
        hole = parent;
    }
    h[hole] = e;
    return top;
}

}  // namespace l3d
