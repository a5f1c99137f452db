// CPU pin of l3d_dev.h: overlap_estimate (the float estimate of the epipolar overlap that k_match_pairs ranks candidates
// by before it spends the double-precision evaluation on them, round 4) against exact_overlap (Line3D::matchingCPU
// line3D.cc:919-958 + mutualOverlap :1086-1165): whenever the estimate says `ok`, |estimate - exact| <= slack -- with
// the exact value 0 where the reference's guards fire.  Same random two-view geometry as prefilter_cover.cpp: targets at
// random, along the source segment's epipolar band, nearly parallel to the pencil (d -> 0), nearly on an epipolar line;
// F scaled over six orders of magnitude.  Also reports how tight the certificate is (largest error against the slack).
//   g++ -std=c++17 -O2 -ffp-contract=off -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tests/cpp/estimate_cover.cpp
#include <cmath>
#include <cstdio>
#include <random>

#include "../../line3dpp_amd/csrc/l3d_dev.h"

using namespace l3d;

int main() {
    std::mt19937_64 rng(2024);
    std::uniform_real_distribution<double> U(-1.0, 1.0), U01(0.0, 1.0);
    unsigned long n = 0, certified = 0, bad = 0; double worst = 0;
    for (int scene = 0; scene < 400; ++scene) {
        // two cameras: K [R | t], fundamental matrix source -> target
        const double W = 600 + 1400 * U01(rng), H = 400 + 1000 * U01(rng), f = (0.6 + 1.2 * U01(rng)) * W;
        const double cx = 0.5 * W, cy = 0.5 * H;
        const double ax = 0.5 * U(rng), ay = 0.5 * U(rng), az = 0.3 * U(rng);
        const double R[9] = {cos(ay) * cos(az), -cos(ay) * sin(az), sin(ay),
                             sin(ax) * sin(ay) * cos(az) + cos(ax) * sin(az), -sin(ax) * sin(ay) * sin(az) + cos(ax) * cos(az), -sin(ax) * cos(ay),
                             -cos(ax) * sin(ay) * cos(az) + sin(ax) * sin(az), cos(ax) * sin(ay) * sin(az) + sin(ax) * cos(az), cos(ax) * cos(ay)};
        double t[3] = {U(rng), U(rng), (scene % 5 == 0 ? 1.0 : 0.2) * U(rng)};
        const double Tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
        const double Ki[9] = {1 / f, 0, -cx / f, 0, 1 / f, -cy / f, 0, 0, 1};
        double TR[9], E[9], F[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double a = 0; for (int k = 0; k < 3; ++k) a += Tx[3 * i + k] * R[3 * k + j]; TR[3 * i + j] = a; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double a = 0; for (int k = 0; k < 3; ++k) a += TR[3 * i + k] * Ki[3 * k + j]; E[3 * i + j] = a; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double a = 0; for (int k = 0; k < 3; ++k) a += Ki[3 * k + i] * E[3 * k + j]; F[3 * i + j] = a; }
        const double fs = std::pow(10.0, 3 * U(rng));   // F is only defined up to scale
        for (double& x : F) x *= fs;
        for (int row = 0; row < 40; ++row) {
            const float sx1 = (float)(W * U01(rng)), sy1 = (float)(H * U01(rng));
            const double len = 5 + 0.4 * W * U01(rng) * U01(rng), ang = M_PI * U(rng);
            const float sx2 = (float)(sx1 + len * cos(ang)), sy2 = (float)(sy1 + len * sin(ang));
            // the kernel's prologue (k_match.hip): the two epipolar lines, normalised, image-centre origin, fp32
            const d3 e1 = mul33(F, d3{(double)sx1, (double)sy1, 1.0}), e2 = mul33(F, d3{(double)sx2, (double)sy2, 1.0});
            const double n1 = sqrt(e1.x * e1.x + e1.y * e1.y), n2 = sqrt(e2.x * e2.x + e2.y * e2.y);
            if (!(n1 > 0.0 && n2 > 0.0)) continue;
            const float e1x = (float)(e1.x / n1), e1y = (float)(e1.y / n1), e1z = (float)((e1.z + (e1.x * cx + e1.y * cy)) / n1);
            const float e2x = (float)(e2.x / n2), e2y = (float)(e2.y / n2), e2z = (float)((e2.z + (e2.x * cx + e2.y * cy)) / n2);
            auto on_line = [&](const d3& e, double px, double py, double& ox, double& oy) {   // foot of (px,py) on line e
                const double nn = e.x * e.x + e.y * e.y, k = (e.x * px + e.y * py + e.z) / nn;
                ox = px - k * e.x; oy = py - k * e.y;
            };
            for (int it = 0; it < 600; ++it) {
                double x1 = W * U01(rng), y1 = H * U01(rng), x2, y2;
                const int kind = it % 6;
                if (kind == 0) { const double l = 5 + 0.4 * W * U01(rng), a = M_PI * U(rng); x2 = x1 + l * cos(a); y2 = y1 + l * sin(a); }
                else if (kind <= 3) {   // end points near the two epipolar lines: partial overlaps of every size
                    double fx, fy, gx, gy;
                    on_line(e1, x1, y1, fx, fy);
                    on_line(e2, x1 + 200 * U(rng), y1 + 200 * U(rng), gx, gy);
                    const double a = 1.5 * U(rng), b = 1.0 + 1.5 * U(rng);   // stretch beyond / inside the band
                    x1 = fx + a * (gx - fx); y1 = fy + a * (gy - fy); x2 = fx + b * (gx - fx); y2 = fy + b * (gy - fy);
                    if (kind == 3) { x2 += 0.5 * U(rng); y2 += 0.5 * U(rng); }
                } else if (kind == 4) { // nearly parallel to the pencil: d -> 0
                    const double l = 5 + 200 * U01(rng), eps = std::pow(10.0, -1 - 7 * U01(rng)) * U(rng);
                    x2 = x1 + l * (-e1.y / n1 + eps * e1.x / n1); y2 = y1 + l * (e1.x / n1 + eps * e1.y / n1);
                } else {                // one end point (nearly) on an epipolar line
                    double fx, fy; on_line(e2, x1, y1, fx, fy);
                    x1 = fx + 1e-3 * U(rng); y1 = fy + 1e-3 * U(rng);
                    const double l = 5 + 300 * U01(rng), a = M_PI * U(rng); x2 = x1 + l * cos(a); y2 = y1 + l * sin(a);
                }
                const float tx1 = (float)x1, ty1 = (float)y1, tx2 = (float)x2, ty2 = (float)y2;
                // SegF of k_prep_views
                const float qx = (float)((double)tx1 - cx), qy = (float)((double)ty1 - cy);
                const float qz = (float)((double)tx1 - (double)tx2), qw = (float)((double)ty1 - (double)ty2);
                const float ov = exact_overlap(F, sx1, sy1, sx2, sy2, tx1, ty1, tx2, ty2);
                float est = 0, slack = 0;
                const bool ok = overlap_estimate(e1x, e1y, e1z, e2x, e2y, e2z, qx, qy, qz, qw, est, slack) && n1 >= 1e-9 && n2 >= 1e-9;
                ++n; certified += ok;
                if (ok) {
                    const double err = fabs((double)est - (double)ov);
                    worst = fmax(worst, err / slack);
                    if (!(err <= slack)) { if (++bad < 10) std::printf("BOUND BROKEN scene %d row %d it %d exact %.9g est %.9g slack %.3g\n", scene, row, it, ov, est, slack); }
                }
            }
        }
    }
    std::printf("%lu pairs, %lu certified, largest error = %.3f of the slack, %lu outside their bound -> %s\n", n, certified, worst, bad,
                bad ? "FAILED" : "covered");
    return bad ? 1 : 0;
}
