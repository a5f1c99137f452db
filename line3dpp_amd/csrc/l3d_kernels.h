// l3d_kernels.h -- launch prototypes shared by the .hip translation units of libl3dpp_hip.so
#pragma once
#include "l3d_dev.h"

namespace l3d {

struct WorkItem {
    uint32_t pair;  // index into the pair array
    uint32_t src0;  // first source segment of the block
};

// ---- k_match.hip ----
size_t match_lds_bytes(int mode, uint32_t K);
hipError_t launch_match_pairs(int mode, bool brute, const ViewDev* views, const PairDesc* pairs,
                              const WorkItem* work, uint32_t nwork, uint32_t maxK, Slot* slots,
                              uint32_t* row_counts, float thr, hipStream_t stream);
hipError_t launch_prep_views(const ViewDev* views, uint32_t n_views, uint32_t max_M, hipStream_t stream);

}  // namespace l3d
