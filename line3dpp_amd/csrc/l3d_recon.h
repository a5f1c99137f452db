// l3d_recon.h -- host-side consumer stages of the hot path (SURVEY.md §8f "next" #1/#2): graph clustering of
// the affinity matrix and the reconstruction tail up to the final 3D segments.  These stages are small and
// sequential in the reference (union-find over weight-sorted edges, one 3x3 eigen problem and one 1-D sweep
// per cluster) and stay on the host here; they exist so that matchImages -> reconstruct3Dlines -> get3Dlines
// is a complete drop-in up to the final 3D segments.
//
//   performClustering            clustering.cc:6-48, universe.h
//   Line3D::clusterSegments      line3D.cc:2079-2152
//   Line3D::get3DlineFromCluster :2155-2218
//   Line3D::project2DsegmentOnto3Dline :2221-2266
//   Line3D::computeFinal3Dsegments :2278-2300, findCollinearSegments(cluster) :2342-2452
//   Line3D::filterTinySegments   :2302-2339, View::projectedLongEnough view.cc:422-427, View::project :374-392
#pragma once
#include <cstdint>
#include <map>
#include <utility>
#include <vector>

#include "l3d_host.h"

namespace l3d {

struct ReconSeg3D {  // L3DPP::Segment3D
    d3 P1{0, 0, 0}, P2{0, 0, 0}, dir{0, 0, 0};
    float length = 0.0f;
    bool valid = false;
};

struct ReconLine {  // L3DPP::FinalLine3D (segment3D.h:165-178)
    std::vector<ReconSeg3D> collinear;                       // collinear3Dsegments_
    ReconSeg3D cluster_seg;                                  // underlyingCluster_.seg3D_
    std::vector<std::pair<uint32_t, uint32_t>> residuals;    // underlyingCluster_.residuals_ (camID, segID)
    uint32_t reference_view = 0;
};

struct ReconView {  // what the tail reads from L3DPP::View (translated frame)
    const HostView* v = nullptr;
};

struct ReconInput {
    // estimated_position3D_ (translated frame) and entry_map_
    std::vector<HypRec> hyps;
    std::map<std::pair<uint32_t, uint32_t>, size_t> entry_map;
    // A_ and local2global_
    std::vector<l3d_cledge> edges;
    std::vector<l3d_segment2d> l2g;
    std::map<uint32_t, const HostView*> views;   // translated
    unsigned visibility_t = 3;
};

// lines3D_ in the translated frame (the caller translates back, line3D.cc:1820 / :559-574)
void reconstruct_lines(const ReconInput& in, std::vector<ReconLine>& out, uint32_t* n_clusters, uint32_t* n_valid);

}  // namespace l3d
