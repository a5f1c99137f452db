// rccl_driver.cpp -- the multi-GPU matchImages (halo form) as a reference-side C++ host would drive it: one process per
// GPU, the C-ABI of libl3dpp_hip.so (include/l3dpp_hip.h) + RCCL, no Python.  The sequence is the one
// line3dpp_amd/dist.py runs over torch.distributed:
//
//   l3d_match_begin                     every rank holds all views (Line3D::addImage on every rank)
//   l3d_plan_shards                     contiguous view ranges with equal pair cost; a rank owns the pairs of its views
//   l3d_match_pairs   (early ranges)    the sub-ranges of its pairs that contain what it has to send
//   l3d_pack_slot_indices               compact index form (4 B per slot) of the pairs whose TARGET view another rank owns
//   ncclGroupStart .. ncclSend/ncclRecv .. ncclGroupEnd      the halo, on a communication stream
//   l3d_match_pairs   (late range)      the rest of its pairs, while the halo travels
//   l3d_expand_slot_indices             received pairs -> slots (re-derived bit-identically)
//   l3d_lists_shard_views               phase B's list pass for the rank's views
//   l3d_shard_options                   round 6: the lowest rank whose records this rank's chain depends on (the ancestors of
//                                       its views in the DAG of pairs that hand inverse matches over)
//   ncclSend/ncclRecv, one group (in place)   the three RECORD arrays of the list pass from a rank to the ranks that depend on
//                                       it, the counter array (overflow flags every rank decides on alike) to everyone
//   world > 1 (round 5: tail and affinity fill sharded by the same views):
//   l3d_tail_shard_count                chain on all records; scores / filterMatches / counts of the rank's views
//                                       (L3D_ERR_RETRY: pools enlarged, repeat the list pass and its exchange)
//   ncclAllGather of the two counts     -> l3d_tail_shard_layout: the rank's outputs at their places, every rank's parts
//   ncclSend/ncclRecv per peer x 9 arrays, one group (in place)   -> l3d_tail_shard_commit
//   l3d_affinity_shard_begin            similarities of the rank's views' surviving matches
//   ncclSend/ncclRecv per peer, one group (in place)              -> l3d_affinity_shard_finish
//   world == 1:  l3d_match_finish, l3d_compute_affinity
//
// usage:  rccl_driver <scene.bin> <rank> <world> <id file>     (scene.bin as tests/cpp/facade_smoke.cpp reads it; rank 0
//         writes the ncclUniqueId to <id file>, the others wait for it; device = rank)
// prints  RESULT rank=.. matches=.. score_sum=.. hypotheses=.. edges=.. rows=.. wsum=..   -- equal on every rank and equal to
//         what a single-GPU l3d_match_images gives (tests compare when a multi-GPU box is at hand; built and link-checked
//         in the CPU container by tests/test_host_logic.py)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "l3dpp_hip.h"

#define L3D(call) do { const int rc_ = (call); if (rc_ != 0) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, l3d_last_error()); return 10; } } while (0)
#define HIP(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 11; } } while (0)
#define NCCL(call) do { const ncclResult_t r_ = (call); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_)); return 12; } } while (0)

struct Run { uint32_t peer, first, count; };

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s scene.bin rank world idfile\n", argv[0]); return 2; }
    const int rank = std::atoi(argv[2]), world = std::atoi(argv[3]);
    HIP(hipSetDevice(rank));
    // ---- communicator ----
    ncclUniqueId id;
    if (rank == 0) {
        NCCL(ncclGetUniqueId(&id));
        FILE* f = std::fopen((std::string(argv[4]) + ".tmp").c_str(), "wb");
        if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) return 3;
        std::fclose(f);
        std::rename((std::string(argv[4]) + ".tmp").c_str(), argv[4]);
    } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 600 && !(f = std::fopen(argv[4], "rb")); ++tries) usleep(100000);
        if (!f || std::fread(&id, sizeof(id), 1, f) != 1) return 3;
        std::fclose(f);
    }
    ncclComm_t comm;
    NCCL(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t comm_stream;
    HIP(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));

    // ---- the scene: every rank adds every view ----
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 4;
    uint32_t nv = 0;
    if (std::fread(&nv, 4, 1, f) != 1) return 4;
    l3d_ctx* c = l3d_create(rank, nullptr);
    if (!c) { std::fprintf(stderr, "l3d_create: %s\n", l3d_last_error()); return 5; }
    std::vector<uint32_t> cams, M;
    for (uint32_t i = 0; i < nv; ++i) {
        uint32_t hdr[5]; double K[9], R[9], t[3]; float md;
        if (std::fread(hdr, 4, 5, f) != 5 || std::fread(K, 8, 9, f) != 9 || std::fread(R, 8, 9, f) != 9 || std::fread(t, 8, 3, f) != 3 ||
            std::fread(&md, 4, 1, f) != 1) return 6;
        std::vector<uint32_t> nb(hdr[4]);
        std::vector<float> segs(4 * (size_t)hdr[1]);
        if (std::fread(nb.data(), 4, hdr[4], f) != hdr[4] || std::fread(segs.data(), 16, hdr[1], f) != hdr[1]) return 6;
        L3D(l3d_add_view(c, hdr[0], segs.data(), hdr[1], K, R, t, hdr[2], hdr[3], md, nb.data(), hdr[4]));
        cams.push_back(hdr[0]); M.push_back(hdr[1]);
    }
    std::fclose(f);
    // view index = rank of the camID (views are kept in ascending camID order)
    std::vector<uint32_t> order(nv);
    for (uint32_t i = 0; i < nv; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cams[a] < cams[b]; });
    auto view_of = [&](uint32_t cam) { return (uint32_t)(std::lower_bound(order.begin(), order.end(), cam, [&](uint32_t i, uint32_t v) { return cams[i] < v; }) - order.begin()); };

    // ---- matchImages, halo form ----
    l3d_match_params prm{2.5f, 10.0f, 10u, 0.25f, 10, -1.0f};
    L3D(l3d_match_begin(c, &prm));
    uint32_t P = 0;
    L3D(l3d_num_pairs(c, &P));
    std::vector<uint32_t> scam(P), tcam(P), sview(P), tview(P);
    std::vector<uint64_t> soff(P + 1), cost(P);
    L3D(l3d_get_pairs(c, scam.data(), tcam.data(), soff.data()));
    for (uint32_t p = 0; p < P; ++p) {
        sview[p] = view_of(scam[p]); tview[p] = view_of(tcam[p]);
        cost[p] = (uint64_t)M[order[sview[p]]] * M[order[tview[p]]];
    }
    std::vector<uint32_t> vb(world + 1), pb(world + 1);
    L3D(l3d_plan_shards(nv, P, sview.data(), cost.data(), (uint32_t)world, vb.data(), pb.data()));
    auto owner = [&](uint32_t v) { return (uint32_t)(std::upper_bound(vb.begin() + 1, vb.end(), v) - (vb.begin() + 1)); };
    std::vector<std::vector<Run>> runs(world);           // runs[r]: what rank r sends (peer, first pair, pair count)
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t r = owner(sview[p]), q = owner(tview[p]);
        if (r == q) continue;
        if (!runs[r].empty() && runs[r].back().peer == q && runs[r].back().first + runs[r].back().count == p) ++runs[r].back().count;
        else runs[r].push_back(Run{q, p, 1});
    }
    // needs[r][q]: rank r's chain depends on the records of rank q (line3dpp_amd/dist.py: shard_needs) -- view v depends on
    // view u < v when a pair (u -> v) exists, transitively; by owner
    std::vector<std::vector<char>> needs(world, std::vector<char>(world, 0));
    {
        std::vector<std::vector<uint32_t>> preds(nv);
        for (uint32_t p = 0; p < P; ++p) if (tview[p] > sview[p]) preds[tview[p]].push_back(sview[p]);
        for (int r = 0; r < world; ++r) {
            std::vector<char> seen(nv, 0);
            std::vector<uint32_t> stack;
            for (uint32_t v = vb[r]; v < vb[r + 1]; ++v) { seen[v] = 1; stack.push_back(v); }
            while (!stack.empty()) {
                const uint32_t v = stack.back(); stack.pop_back();
                for (uint32_t u : preds[v]) if (!seen[u]) { seen[u] = 1; stack.push_back(u); }
            }
            for (uint32_t v = 0; v < nv; ++v) if (seen[v] && (int)owner(v) != r) needs[r][owner(v)] = 1;
        }
    }
    uint32_t first_needed = (uint32_t)rank;
    for (int q = 0; q < rank; ++q) if (needs[rank][q]) { first_needed = (uint32_t)q; break; }
    // (the exchanges below run on a communication stream of their own: the library's entries must return with their parts
    // complete -- exchanges_stream_ordered = 0)
    L3D(l3d_shard_options(c, first_needed, 0));
    const uint32_t first = pb[rank], count = pb[rank + 1] - pb[rank];
    // the largest stretch of own pairs without anything to send is matched last
    uint32_t late_first = first, late_count = count;
    if (!runs[rank].empty()) {
        std::vector<int64_t> edges{(int64_t)first - 1};
        for (const Run& r : runs[rank]) for (uint32_t p = r.first; p < r.first + r.count; ++p) edges.push_back(p);
        edges.push_back((int64_t)first + count);
        late_count = 0;
        for (size_t i = 0; i + 1 < edges.size(); ++i)
            if (edges[i + 1] - edges[i] - 1 > (int64_t)late_count) { late_count = (uint32_t)(edges[i + 1] - edges[i] - 1); late_first = (uint32_t)(edges[i] + 1); }
    }
    if (late_first > first) L3D(l3d_match_pairs(c, first, late_first - first));
    if (late_first + late_count < first + count) L3D(l3d_match_pairs(c, late_first + late_count, first + count - late_first - late_count));
    for (const Run& r : runs[rank]) L3D(l3d_pack_slot_indices(c, r.first, r.count));   // (returns when the indices are there)
    void* idx = nullptr; uint64_t n_slots = 0;
    L3D(l3d_slot_index_buffer(c, &idx, &n_slots));
    soff[P] = n_slots;
    NCCL(ncclGroupStart());
    for (int r = 0; r < world; ++r)
        for (const Run& run : runs[r]) {
            uint32_t* at = (uint32_t*)idx + soff[run.first];
            const size_t n = (size_t)(soff[run.first + run.count] - soff[run.first]);
            if (r == rank) NCCL(ncclSend(at, n, ncclUint32, (int)run.peer, comm, comm_stream));
            else if ((int)run.peer == rank) NCCL(ncclRecv(at, n, ncclUint32, r, comm, comm_stream));
        }
    NCCL(ncclGroupEnd());
    if (late_count) L3D(l3d_match_pairs(c, late_first, late_count));       // the GPU matches while the halo travels
    HIP(hipStreamSynchronize(comm_stream));
    for (int r = 0; r < world; ++r)
        for (const Run& run : runs[r])
            if ((int)run.peer == rank) L3D(l3d_expand_slot_indices(c, run.first, run.count));
    int rc = L3D_ERR_RETRY;
    for (int attempt = 0; attempt < 8 && rc == L3D_ERR_RETRY; ++attempt) {
        void* slab[4]; uint64_t bytes[4]; void* full[4];
        L3D(l3d_lists_shard_views(c, (uint32_t)rank, (uint32_t)world, vb[rank], vb[rank + 1], slab, bytes, full));
        // direct exchange, in place (the rank's slab is its own slice of the array): one send and one receive per peer
        // and array in ONE group -- the xGMI mesh gives every pair of GPUs its own link, so the N-1 transfers of a rank
        // run side by side (a ring all-gather would push N-1 slabs through one link)
        NCCL(ncclGroupStart());
        for (int k = 0; k < 4; ++k)
            for (int q = 0; q < world; ++q) {
                if (q == rank || !bytes[k]) continue;
                // records (k < 3): to the ranks that depend on them; counters (k = 3): to everyone
                if (k == 3 || needs[q][rank]) NCCL(ncclSend(slab[k], (size_t)bytes[k], ncclUint8, q, comm, comm_stream));
                if (k == 3 || needs[rank][q]) NCCL(ncclRecv((char*)full[k] + (size_t)q * bytes[k], (size_t)bytes[k], ncclUint8, q, comm, comm_stream));
            }
        NCCL(ncclGroupEnd());
        HIP(hipStreamSynchronize(comm_stream));
        if (world == 1) { rc = l3d_match_finish(c); continue; }
        // ---- the tail sharded by views: counts -> layout -> parts in place -> commit ----
        uint32_t mine[2] = {0, 0};
        rc = l3d_tail_shard_count(c, mine);
        if (rc != 0) continue;                                   // (L3D_ERR_RETRY on every rank alike; anything else ends the loop)
        uint32_t *d_cnt = nullptr;
        HIP(hipMalloc(&d_cnt, (size_t)(2 + 2 * world) * 4));
        HIP(hipMemcpy(d_cnt, mine, 8, hipMemcpyHostToDevice));
        NCCL(ncclAllGather(d_cnt, d_cnt + 2, 2, ncclUint32, comm, comm_stream));
        HIP(hipStreamSynchronize(comm_stream));
        std::vector<uint32_t> counts(2 * (size_t)world);
        HIP(hipMemcpy(counts.data(), d_cnt + 2, counts.size() * 4, hipMemcpyDeviceToHost));
        HIP(hipFree(d_cnt));
        void* base[9]; uint64_t elt[9];
        std::vector<uint64_t> pfirst(9 * (size_t)world), pcount(9 * (size_t)world);
        L3D(l3d_tail_shard_layout(c, (uint32_t)world, counts.data(), vb.data(), base, elt, pfirst.data(), pcount.data()));
        NCCL(ncclGroupStart());
        for (int k = 0; k < 9; ++k)
            for (int q = 0; q < world; ++q) {
                if (q == rank) continue;
                const uint64_t fq = pfirst[9 * q + k], nq = pcount[9 * q + k], fm = pfirst[9 * rank + k], nm = pcount[9 * rank + k];
                if (nq) NCCL(ncclRecv((char*)base[k] + fq * elt[k], (size_t)(nq * elt[k]), ncclUint8, q, comm, comm_stream));
                if (nm) NCCL(ncclSend((char*)base[k] + fm * elt[k], (size_t)(nm * elt[k]), ncclUint8, q, comm, comm_stream));
            }
        NCCL(ncclGroupEnd());
        HIP(hipStreamSynchronize(comm_stream));
        rc = l3d_tail_shard_commit(c);
    }
    if (rc != 0) { std::fprintf(stderr, "phase B -> %d: %s\n", rc, l3d_last_error()); return 13; }
    if (world == 1) L3D(l3d_compute_affinity(c));
    else {
        void* simv = nullptr;
        std::vector<uint64_t> afirst(world), acount(world);
        L3D(l3d_affinity_shard_begin(c, (uint32_t)rank, (uint32_t)world, &simv, afirst.data(), acount.data()));
        HIP(hipDeviceSynchronize());                             // the similarities of this rank's views are written
        NCCL(ncclGroupStart());
        for (int q = 0; q < world; ++q) {
            if (q == rank) continue;
            if (acount[q]) NCCL(ncclRecv((float*)simv + afirst[q], (size_t)acount[q], ncclFloat, q, comm, comm_stream));
            if (acount[rank]) NCCL(ncclSend((float*)simv + afirst[rank], (size_t)acount[rank], ncclFloat, q, comm, comm_stream));
        }
        NCCL(ncclGroupEnd());
        HIP(hipStreamSynchronize(comm_stream));
        L3D(l3d_affinity_shard_finish(c));
    }

    // ---- what every rank must hold: the complete result ----
    uint64_t n_matches = 0; double score_sum = 0.0;
    for (uint32_t i = 0; i < nv; ++i) {
        uint64_t n = 0;
        std::vector<uint32_t> offs(M[i] + 1);
        L3D(l3d_get_matches(c, cams[i], nullptr, 0, offs.data(), &n));
        std::vector<l3d_match> m(std::max<uint64_t>(n, 1));
        L3D(l3d_get_matches(c, cams[i], m.data(), n, offs.data(), &n));
        n_matches += n;
        for (uint64_t k = 0; k < n; ++k) score_sum += m[k].score3D_;
    }
    uint32_t n_best = 0, n_edges = 0, n_rows = 0;
    L3D(l3d_num_best(c, &n_best));
    L3D(l3d_num_affinity(c, &n_edges, &n_rows));
    std::vector<l3d_cledge> e(std::max<uint32_t>(n_edges, 1)); std::vector<l3d_segment2d> l2g(std::max<uint32_t>(n_rows, 1)); float msdl = 0;
    L3D(l3d_get_affinity(c, e.data(), l2g.data(), &msdl));
    double wsum = 0.0;
    for (uint32_t k = 0; k < n_edges; ++k) wsum += e[k].w_;
    std::printf("RESULT rank=%d matches=%llu score_sum=%.6f hypotheses=%u edges=%u rows=%u wsum=%.6f\n", rank,
                (unsigned long long)n_matches, score_sum, n_best, n_edges, n_rows, wsum);
    l3d_destroy(c);
    ncclCommDestroy(comm);
    return 0;
}
