// l3d_neighbors.hip -- visual neighbours from shared worldpoints: Line3D::findVisualNeighborsFromWPs
// (line3D.cc:578-699) with processWPlist (:230-240) and the View helpers it calls (opticalAxesAngle view.cc:457-463,
// distanceVisualNeighborScore :487-501, baseLine :504-507).  Host code: the caller side of the matching path for
// users who hand over SfM worldpoint lists (main_vsfm.cpp, main_colmap.cpp ...) instead of explicit neighbour lists.
// Runs inside l3d_match_begin on the translated views, exactly where matchImages calls it (:480-484), for every view
// that was added with worldpoints -- every call anew, as the reference does.
#include <list>
#include <map>
#include <set>

#include "l3d_ctx.h"

namespace l3d {

namespace {

struct VisualNeighbor {   // commons.h:160-166
    uint32_t cam;
    float score, axis_angle, distance_score;
};

// View::getNormalizedRay(pp_), view.cc:317-321 / :451-454
d3 optical_axis(const HostView& v) { return normalized(mul33(v.RtKinv.m, v.pp)); }

// View::opticalAxesAngle, view.cc:457-463
double optical_axes_angle(const HostView& v, const HostView& o) {
    return std::acos(std::fmin(std::fmax(dot(optical_axis(v), optical_axis(o)), -1.0), 1.0));
}
// View::distanceVisualNeighborScore, view.cc:487-501: |x| + |y| of the other centre in this camera's frame, in float
float distance_score(const HostView& v, const HostView& o) {
    const d3 c = mul33(v.R.m, o.C) + v.t;
    const float d1 = (float)std::fabs(c.x), d2 = (float)std::fabs(c.y);
    return d1 + d2;
}
// View::baseLine, view.cc:504-507
float base_line(const HostView& v, const HostView& o) { return (float)norm(v.C - o.C); }

}  // namespace

// visual_nbrs of every view of `views` that carries worldpoints; `views` are the context's views (translated)
void neighbors_from_worldpoints(const std::map<uint32_t, HostView*>& views, uint32_t num_neighbors) {
    // worldpoints2views_ / num_worldpoints_, processWPlist
    std::map<uint32_t, std::list<uint32_t>> wp2views;
    for (const auto& kv : views)
        if (kv.second->by_worldpoints)
            for (uint32_t wp : kv.second->worldpoints) wp2views[wp].push_back(kv.first);
    for (const auto& kv : views) {
        HostView* v = kv.second;
        if (!v->by_worldpoints) continue;
        const uint32_t cam = kv.first;
        v->visual_nbrs.clear();                                   // :582-583 reset
        std::map<uint32_t, uint32_t> common;                      // view -> number of shared worldpoints
        for (uint32_t wp : v->worldpoints)
            for (uint32_t o : wp2views[wp])
                if (o != cam) ++common[o];
        if (common.empty()) continue;
        std::list<VisualNeighbor> nb;
        for (const auto& c : common) {                            // ascending camera id
            const HostView* o = views.at(c.first);
            VisualNeighbor vn;
            vn.cam = c.first;
            vn.score = 2.0f * float(c.second) / float((uint32_t)(v->worldpoints.size() + o->worldpoints.size()));
            vn.axis_angle = (float)optical_axes_angle(*v, *o);
            vn.distance_score = distance_score(*v, *o);
            if (vn.axis_angle < 1.571f && c.second > 4) nb.push_back(vn);
        }
        nb.sort([](const VisualNeighbor a, const VisualNeighbor b) { return a.score > b.score; });   // stable, :637
        if (nb.size() > num_neighbors) {                          // :640-666
            std::list<VisualNeighbor> all = nb;
            const float score_t = 0.80f * nb.front().score;
            uint32_t n_big = 0;
            for (auto it = nb.begin(); it != nb.end() && it->score > score_t; ++it) ++n_big;
            nb.resize(n_big);
            nb.sort([](const VisualNeighbor a, const VisualNeighbor b) { return a.distance_score > b.distance_score; });
            if (nb.size() > num_neighbors / 2) nb.resize(num_neighbors / 2);
            nb.splice(nb.end(), all);
        }
        const float min_baseline = 0.1f;                          // :669-670 (the computed value is overwritten there)
        std::set<uint32_t> used;
        for (auto it = nb.begin(); it != nb.end() && used.size() < num_neighbors; ++it) {
            const HostView* o = views.at(it->cam);
            if (!used.count(it->cam) && base_line(*v, *o) > min_baseline) {
                bool valid = true;                                // :680-686: the reference tests v against the neighbours
                for (uint32_t u : used) {                         // taken so far (not the candidate against them)
                    if (!(base_line(*v, *views.at(u)) > min_baseline)) { valid = false; break; }
                }
                if (valid) used.insert(it->cam);
            }
        }
        v->visual_nbrs = used;
    }
}

}  // namespace l3d

// Context-free form for callers that only want the neighbour lists (and for the host-logic tests): the cameras as
// they are handed to addImage; translated and moved back here like matchImages does around its own call.
extern "C" int l3d_neighbors_from_worldpoints(uint32_t n_views, const uint32_t* cam_ids, const double* K9,
                                              const double* R9, const double* t3, const uint64_t* wp_offsets,
                                              const uint32_t* worldpoints, uint32_t num_neighbors,
                                              uint64_t* nb_offsets, uint32_t* neighbors, uint64_t cap) {
    if (!n_views || !cam_ids || !K9 || !R9 || !t3 || !wp_offsets || !nb_offsets) return fail(L3D_ERR_ARG, "null argument");
    if (wp_offsets[n_views] && !worldpoints) return fail(L3D_ERR_ARG, "null argument");
    std::vector<std::unique_ptr<HostView>> store;
    std::map<uint32_t, HostView*> views;
    for (uint32_t i = 0; i < n_views; ++i) {
        if (views.count(cam_ids[i])) return fail(L3D_ERR_ID_IN_USE, "camera ID already in use");
        auto v = std::make_unique<HostView>();
        v->cam = cam_ids[i];
        init_view(*v, K9 + 9 * (size_t)i, R9 + 9 * (size_t)i, t3 + 3 * (size_t)i);
        v->by_worldpoints = true;
        v->worldpoints.assign(worldpoints + wp_offsets[i], worldpoints + wp_offsets[i + 1]);
        views[v->cam] = v.get();
        store.push_back(std::move(v));
    }
    std::vector<HostView*> order;
    for (auto& kv : views) order.push_back(kv.second);
    const d3 tr = scene_translation(order);
    for (auto* v : order) translate_view(*v, d3{-tr.x, -tr.y, -tr.z});
    neighbors_from_worldpoints(views, std::max(num_neighbors, 2u));            // clamp of matchImages, line3D.cc:396
    uint64_t n = 0;
    for (uint32_t i = 0; i < n_views; ++i) {
        nb_offsets[i] = n;
        for (uint32_t o : views[cam_ids[i]]->visual_nbrs) {
            if (neighbors && n < cap) neighbors[n] = o;
            ++n;
        }
    }
    nb_offsets[n_views] = n;
    return (neighbors && n > cap) ? fail(L3D_ERR_LIMIT, "neighbour buffer too small") : L3D_OK;
}
