// CPU pin of line3dpp_amd/csrc/l3d_heap.h against std::priority_queue (the container Line3D::matchingCPU uses for
// its kNN selection, line3D.cc:931-1007 / commons.h:217-231): tie-heavy random sequences, every length up to 300.
#include <cstdio>
#include <queue>
#include <random>
#include <vector>

#include "../../line3dpp_amd/csrc/l3d_heap.h"

struct M { float overlap; unsigned tgt; };
struct Cmp { bool operator()(const M& a, const M& b) const { return a.overlap < b.overlap; } };

int main() {
    std::mt19937 rng(1234);
    unsigned long checked = 0;
    for (int levels : {1, 2, 3, 5, 17, 1000000})
        for (unsigned n = 0; n <= 300; ++n) {
            std::priority_queue<M, std::vector<M>, Cmp> pq;
            std::vector<float> ov(n + 1); std::vector<unsigned> ix(n + 1);
            std::vector<uint64_t> packed(n + 1);
            for (unsigned c = 0; c < n; ++c) {
                const float o = 0.25f + 0.7f * (float)(rng() % levels) / (float)levels;
                pq.push(M{o, c});
                l3d::heap_push(ov.data(), ix.data(), c, o, c);
                l3d::heap_push_packed(packed.data(), c, l3d::heap_pack(o, c));
            }
            for (unsigned left = n; left > 0; --left) {
                const M t = pq.top(); pq.pop();
                float v; unsigned x;
                l3d::heap_pop(ov.data(), ix.data(), left, v, x);
                const uint64_t e = l3d::heap_pop_packed(packed.data(), left);
                if (l3d::heap_overlap(e) != t.overlap || (unsigned)e != t.tgt) {
                    std::printf("packed mismatch: levels %d n %u left %u\n", levels, n, left);
                    return 1;
                }
                if (v != t.overlap || x != t.tgt) {
                    std::printf("mismatch: levels %d n %u left %u: (%g,%u) vs (%g,%u)\n", levels, n, left, v, x, t.overlap, t.tgt);
                    return 1;
                }
                ++checked;
            }
        }
    std::printf("heap order identical to std::priority_queue on %lu pops\n", checked);
    return 0;
}
