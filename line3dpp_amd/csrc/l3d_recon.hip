// l3d_recon.hip -- graph clustering + reconstruction tail on the host (see l3d_recon.h for the reference map).
#include "l3d_recon.h"

#include <algorithm>
#include <cmath>

namespace l3d {

namespace {

// L3DPP::Segment3D(P1,P2), segment3D.h:47-66
ReconSeg3D make_seg(const d3& a, const d3& b) {
    ReconSeg3D s;
    s.length = (float)norm(a - b);
    if (s.length > kEps) {
        s.P1 = a; s.P2 = b; s.dir = normalized(b - a); s.valid = true;
    } else {
        s.length = 0.0f; s.valid = false;
    }
    return s;
}

// CLUniverse, universe.h: union-find with rank, path "compression" of the queried node only
struct Universe {
    struct E { int rank, cluster, size; };
    std::vector<E> e;
    explicit Universe(int n) : e(n) { for (int i = 0; i < n; ++i) e[i] = E{0, i, 1}; }
    int find(int x) {
        int y = x;
        while (y != e[y].cluster) y = e[y].cluster;
        e[x].cluster = y;
        return y;
    }
    void join(int x, int y) {
        if (e[x].rank > e[y].rank) { e[y].cluster = x; e[x].size += e[y].size; }
        else { e[x].cluster = y; e[y].size += e[x].size; if (e[x].rank == e[y].rank) e[y].rank++; }
    }
};

// performClustering, clustering.cc:6-48 (c = 3.0, line3D.cc:2089).  std::list::sort is a stable merge sort.
Universe perform_clustering(std::vector<l3d_cledge> edges, int num_nodes, float c) {
    std::stable_sort(edges.begin(), edges.end(), [](const l3d_cledge& a, const l3d_cledge& b) { return a.w_ < b.w_; });
    Universe u(num_nodes);
    std::vector<float> threshold(num_nodes, c);
    for (const l3d_cledge& e : edges) {
        int a = u.find(e.i_), b = u.find(e.j_);
        if (a != b && e.w_ <= threshold[a] && e.w_ <= threshold[b]) {
            u.join(a, b);
            a = u.find(a);
            threshold[a] = e.w_ + c / (float)u.e[a].size;
        }
    }
    return u;
}

// Principal direction of a symmetric positive semi-definite 3x3 matrix -- what JacobiSVD + maxCoeff give in
// get3DlineFromCluster (line3D.cc:2196-2211).  Not an iteration of Jacobi rotations (that is how the checker's Eigen
// stand-in, oracle/ref_shim, solves it; the product must not be compared with its own code): the largest eigenvalue in
// closed form (trigonometric solution of the characteristic cubic of the trace-free, scaled matrix), its eigenvector as
// the largest cross product of two rows of S - lambda*I (the rows span the plane orthogonal to it), polished by two
// steps of power iteration with the matrix itself.  The sign of the direction is free (the sweep of
// findCollinearSegments starts from the end point farthest from the centre of gravity, whichever way the line points);
// the largest component is made positive.
d3 principal_direction(double S[3][3]) {
    double scale = 0.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) scale = std::fmax(scale, std::fabs(S[i][j]));
    if (!(scale > 0.0) || !std::isfinite(scale)) return d3{1.0, 0.0, 0.0};
    double A[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = 0.5 * (S[i][j] + S[j][i]) / scale;
    const double q = (A[0][0] + A[1][1] + A[2][2]) / 3.0;
    const double p1 = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double d0 = A[0][0] - q, d1 = A[1][1] - q, d2 = A[2][2] - q;
    const double p2 = d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * p1;
    d3 v{1.0, 0.0, 0.0};
    if (p2 > 1e-300) {
        const double p = std::sqrt(p2 / 6.0);
        // B = (A - q I) / p; det(B) / 2 = cos(3 phi)
        const double b00 = d0 / p, b11 = d1 / p, b22 = d2 / p, b01 = A[0][1] / p, b02 = A[0][2] / p, b12 = A[1][2] / p;
        const double detB = b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02);
        const double r = std::fmin(std::fmax(0.5 * detB, -1.0), 1.0);
        const double lambda = q + 2.0 * p * std::cos(std::acos(r) / 3.0);     // the largest of the three roots
        const d3 r0{A[0][0] - lambda, A[0][1], A[0][2]}, r1{A[0][1], A[1][1] - lambda, A[1][2]}, r2{A[0][2], A[1][2], A[2][2] - lambda};
        const d3 c[3] = {cross(r0, r1), cross(r0, r2), cross(r1, r2)};
        int best = 0; double bn = -1.0;
        for (int k = 0; k < 3; ++k) { const double n2 = dot(c[k], c[k]); if (n2 > bn) { bn = n2; best = k; } }
        if (bn > 1e-300) v = c[best] * (1.0 / std::sqrt(bn));
        else {   // two equal largest eigenvalues or a multiple of the identity: any direction of the eigenspace; take
                 // the coordinate axis with the largest diagonal entry as the start of the power iteration
            const int m = (A[1][1] > A[0][0] ? (A[2][2] > A[1][1] ? 2 : 1) : (A[2][2] > A[0][0] ? 2 : 0));
            v = d3{m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0, m == 2 ? 1.0 : 0.0};
        }
    }
    for (int it = 0; it < 2; ++it) {
        const d3 w{A[0][0] * v.x + A[0][1] * v.y + A[0][2] * v.z, A[0][1] * v.x + A[1][1] * v.y + A[1][2] * v.z,
                   A[0][2] * v.x + A[1][2] * v.y + A[2][2] * v.z};
        const double n = norm(w);
        if (!(n > 1e-300)) break;
        v = w * (1.0 / n);
    }
    const double ax = std::fabs(v.x), ay = std::fabs(v.y), az = std::fabs(v.z);
    const double lead = (ax >= ay && ax >= az) ? v.x : (ay >= az ? v.y : v.z);
    if (lead < 0.0) v = v * -1.0;
    return normalized(v);
}

d3 ray_of(const HostView& v, uint32_t seg, bool first) {  // View::getNormalizedLinePointRay, view.cc:330-353
    if (seg >= v.M) return d3{0, 0, 0};
    const float* s = &v.segs[4 * (size_t)seg];
    const d3 p = first ? d3{(double)s[0], (double)s[1], 1.0} : d3{(double)s[2], (double)s[3], 1.0};
    return normalized(mul33(v.RtKinv.m, p));
}

// Line3D::project2DsegmentOnto3Dline, line3D.cc:2221-2266
bool project_onto_line(const HostView& v, uint32_t seg, const ReconSeg3D& L, ReconSeg3D& out) {
    const d3 P = L.P1, u = L.dir, Q = v.C;
    const d3 v1 = ray_of(v, seg, true), v2 = ray_of(v, seg, false);
    const d3 w = P - Q;
    const double a = dot(u, u), b1 = dot(u, v1), b2 = dot(u, v2), c1 = dot(v1, v1), c2 = dot(v2, v2);
    const double d = dot(u, w), e1 = dot(v1, w), e2 = dot(v2, w);
    const double denom1 = a * c1 - b1 * b1, denom2 = a * c2 - b2 * b2;
    if (std::fabs(denom1) > kEps && std::fabs(denom2) > kEps) {
        const double s1 = (b1 * e1 - c1 * d) / denom1, s2 = (b2 * e2 - c2 * d) / denom2;
        out = make_seg(P + u * s1, P + u * s2);
        return true;
    }
    out = ReconSeg3D();
    return false;
}

// View::project, view.cc:374-392
void project(const HostView& v, const d3& P, double& x, double& y) {
    const d3 rp = mul33(v.R.m, P);
    d3 q = rp + v.t;
    const double xn = (1.0 * q.x + 0.0 * q.z) / q.z, yn = (1.0 * q.y + 0.0 * q.z) / q.z;
    q = mul33(v.K.m, d3{xn, yn, 1.0});
    x = q.x / q.z; y = q.y / q.z;
}

}  // namespace

d3 principal_direction_of(double S[3][3]) { return principal_direction(S); }   // (named entry for the test hook below)

void reconstruct_lines(const ReconInput& in, std::vector<ReconLine>& out, uint32_t* n_clusters, uint32_t* n_valid) {
    out.clear();
    if (n_clusters) *n_clusters = 0;
    if (n_valid) *n_valid = 0;
    if (in.edges.empty()) return;
    // ---- clusterSegments, line3D.cc:2079-2152 ----
    Universe u = perform_clustering(in.edges, (int)in.l2g.size(), 3.0f);
    std::map<int, std::vector<std::pair<uint32_t, uint32_t>>> cluster2segments;
    std::map<int, std::map<uint32_t, bool>> cluster2cameras;
    std::vector<int> unique_clusters;
    for (size_t id = 0; id < in.l2g.size(); ++id) {
        const int cl = u.find((int)id);
        if (cluster2segments.find(cl) == cluster2segments.end()) unique_clusters.push_back(cl);
        cluster2segments[cl].push_back({in.l2g[id].camID_, in.l2g[id].segID_});
        cluster2cameras[cl][in.l2g[id].camID_] = true;
    }
    if (n_clusters) *n_clusters = (uint32_t)cluster2segments.size();
    struct Cluster { ReconSeg3D seg; std::vector<std::pair<uint32_t, uint32_t>> residuals; uint32_t ref_view; };
    std::vector<Cluster> clusters;
    for (int cl : unique_clusters) {
        if (cluster2cameras[cl].size() < in.visibility_t) continue;
        // ---- get3DlineFromCluster, :2155-2218 ----
        const auto& segs = cluster2segments[cl];
        const int n = (int)segs.size() * 2;
        std::vector<d3> pts;
        pts.reserve(n);
        d3 P{0, 0, 0};
        uint32_t reference_cam = 0;
        float max_len_2D = 0.0f;
        for (const auto& s2 : segs) {
            const HypRec& h = in.hyps[in.entry_map.at(s2)];
            const d3 a{h.P1[0], h.P1[1], h.P1[2]}, b{h.P2[0], h.P2[1], h.P2[2]};
            P = P + a; P = P + b;
            pts.push_back(a); pts.push_back(b);
            const HostView& v = *in.views.at(s2.first);
            const float* c = &v.segs[4 * (size_t)s2.second];
            const float length_sqr = (c[0] - c[2]) * (c[0] - c[2]) + (c[1] - c[3]) * (c[1] - c[3]);
            if (length_sqr > max_len_2D) { max_len_2D = length_sqr; reference_cam = s2.first; }
        }
        P = d3{P.x / double(n), P.y / double(n), P.z / double(n)};
        // scatter matrix L*(I - 11^T/n)*L^T = sum of outer products of the centred points
        d3 mean{0, 0, 0};
        for (const d3& p : pts) mean = mean + p;
        mean = d3{mean.x / n, mean.y / n, mean.z / n};
        double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (const d3& p : pts) {
            const double c[3] = {p.x - mean.x, p.y - mean.y, p.z - mean.z};
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) S[i][j] += c[i] * c[j];
        }
        const d3 dir = principal_direction(S);
        Cluster c;
        c.seg = make_seg(P - dir, P + dir);
        c.residuals = segs;
        c.ref_view = reference_cam;
        if (!c.residuals.empty()) clusters.push_back(std::move(c));
    }
    if (n_valid) *n_valid = (uint32_t)clusters.size();
    // ---- computeFinal3Dsegments :2278-2300 + findCollinearSegments(cluster) :2342-2452 ----
    for (const Cluster& cl : clusters) {
        std::vector<ReconSeg3D> collinear;
        const d3 COG = (cl.seg.P1 + cl.seg.P2) * 0.5;
        struct Pt { size_t line, point; uint32_t cam; float dist; };
        std::vector<Pt> line_points;
        std::vector<d3> pts(cl.residuals.size() * 2);
        float dist_to_cog = 0.0f;
        d3 border{0, 0, 0};
        size_t pID = 0;
        for (size_t id = 0; id < cl.residuals.size(); ++id, pID += 2) {
            ReconSeg3D proj;
            if (!project_onto_line(*in.views.at(cl.residuals[id].first), cl.residuals[id].second, cl.seg, proj)) continue;
            pts[pID] = proj.P1; pts[pID + 1] = proj.P2;
            line_points.push_back(Pt{id, pID, cl.residuals[id].first, 0.0f});
            float d = (float)norm(proj.P1 - COG);
            if (d > dist_to_cog) { dist_to_cog = d; border = proj.P1; }
            line_points.push_back(Pt{id, pID + 1, cl.residuals[id].first, 0.0f});
            d = (float)norm(proj.P2 - COG);
            if (d > dist_to_cog) { dist_to_cog = d; border = proj.P2; }
        }
        if (line_points.size() >= 6) {
            for (Pt& p : line_points) p.dist = (float)norm(pts[p.point] - border);
            std::stable_sort(line_points.begin(), line_points.end(), [](const Pt& a, const Pt& b) { return a.dist < b.dist; });
            std::map<size_t, unsigned> open;       // camera -> number of open 2D segments
            std::map<size_t, bool> open_lines;
            bool opened = false;
            d3 current_start{0, 0, 0};
            for (const Pt& pt : line_points) {
                if (open_lines.find(pt.line) == open_lines.end()) {
                    open_lines[pt.line] = true;
                    ++open[pt.cam];
                } else {
                    open_lines.erase(pt.line);
                    if (--open[pt.cam] == 0) open.erase(pt.cam);
                }
                if (opened && open.size() < 3) {
                    collinear.push_back(make_seg(current_start, pts[pt.point]));
                    opened = false;
                } else if (!opened && open.size() >= 3) {
                    current_start = pts[pt.point];
                    opened = true;
                }
            }
        }
        if (collinear.empty()) continue;
        // ---- filterTinySegments, :2302-2339 ----
        const HostView& rv = *in.views.at(cl.ref_view);
        const float diagonal = std::sqrt(float(rv.width * rv.width + rv.height * rv.height));
        const float min_line_length = diagonal * 0.005f;   // L3D_DEF_MIN_LINE_LENGTH_FACTOR, view.cc:17-18
        ReconLine L;
        for (const ReconSeg3D& s : collinear) {
            double x1, y1, x2, y2;
            project(rv, s.P1, x1, y1); project(rv, s.P2, x2, y2);
            const double dx = x1 - x2, dy = y1 - y2;
            if (std::sqrt(dx * dx + dy * dy) > (double)min_line_length) L.collinear.push_back(s);
        }
        if (L.collinear.empty()) continue;
        L.cluster_seg = cl.seg;
        L.residuals = cl.residuals;
        L.reference_view = cl.ref_view;
        out.push_back(std::move(L));
    }
}

}  // namespace l3d

// test hook (host only, no device needed): the principal direction get3DlineFromCluster takes from its 3x3 scatter
// matrix (row-major S9), as this library computes it -- checked against LAPACK in tests/test_host_logic.py
extern "C" int l3d_principal_direction(const double S9[9], double dir3[3]) {
    if (!S9 || !dir3) return -1;
    double S[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) S[i][j] = S9[3 * i + j];
    const l3d::d3 d = l3d::principal_direction_of(S);
    dir3[0] = d.x; dir3[1] = d.y; dir3[2] = d.z;
    return 0;
}
