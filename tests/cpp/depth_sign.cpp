// CPU pin of l3d_dev.h: depths_positive (the decision of the depth test with one multiplication per depth instead of an
// IEEE division, used by the first stage of k_match_pairs' candidate pipeline) against exact_depths (the reference's
// arithmetic, Line3D::triangulationDepths line3D.cc:1168-1193 + the acceptance test :966-980) -- random geometry,
// depths forced to the neighbourhood of L3D_EPS (where the two roundings could disagree and the fallback must take
// over), degenerate denominators, NaN / inf.
//   g++ -std=c++17 -O2 -ffp-contract=off -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tests/cpp/depth_sign.cpp
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../line3dpp_amd/csrc/l3d_dev.h"

using namespace l3d;

int main() {
    std::mt19937_64 rng(99);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    unsigned long n = 0, accepted = 0, sliver = 0, bad = 0;
    auto unit = [&](double* v) { double x = U(rng), y = U(rng), z = U(rng), l = std::sqrt(x * x + y * y + z * z) + 1e-30; v[0] = x / l; v[1] = y / l; v[2] = z / l; };
    for (int mode = 0; mode < 6; ++mode)
        for (int it = 0; it < 400000; ++it) {
            SegX sx, tx; double Cs[3], Ct[3];
            unit(sx.r1); unit(sx.r2); unit(sx.n); unit(tx.r1); unit(tx.r2); unit(tx.n); unit(sx.rm); unit(tx.rm);
            for (int k = 0; k < 3; ++k) { Cs[k] = 10 * U(rng); Ct[k] = 10 * U(rng); }
            sx.cn = Cs[0] * sx.n[0] + (Cs[1] * sx.n[1] + Cs[2] * sx.n[2]);
            tx.cn = Ct[0] * tx.n[0] + (Ct[1] * tx.n[1] + Ct[2] * tx.n[2]);
            if (mode == 1) {          // num / da right at L3D_EPS: cn chosen so that the first depth is eps * (1 + k ulp)
                const double da = sx.r1[0] * tx.n[0] + (sx.r1[1] * tx.n[1] + sx.r1[2] * tx.n[2]);
                const double nc = tx.n[0] * Cs[0] + (tx.n[1] * Cs[1] + tx.n[2] * Cs[2]);
                double target = 1e-12 * da;
                for (int k = (int)(rng() % 9) - 4; k != 0; k += k > 0 ? -1 : 1) target = std::nextafter(target, k > 0 ? 1e300 : -1e300);
                tx.cn = target + nc;
            }
            if (mode == 2) { tx.n[0] = 1e-13 * U(rng); tx.n[1] = 1e-13 * U(rng); tx.n[2] = 1e-13 * U(rng); }   // |da| < eps
            if (mode == 3) { const double s = std::pow(10.0, 12 * U(rng)); for (int k = 0; k < 3; ++k) { Cs[k] *= s; } tx.cn *= s; }
            if (mode == 4 && (it % 7) == 0) tx.cn = NAN;
            if (mode == 4 && (it % 7) == 1) tx.cn = INFINITY;
            if (mode == 5) { for (int k = 0; k < 3; ++k) Ct[k] = Cs[k] + 1e-9 * U(rng); tx.cn = Ct[0] * tx.n[0] + (Ct[1] * tx.n[1] + Ct[2] * tx.n[2]); }
            PairResult res{};
            const bool want = exact_depths(sx, tx, Cs, Ct, res);
            const SegD& td = *reinterpret_cast<const SegD*>(&tx);           // the match kernel reads targets as SegD
            const bool got = depths_positive(sx, td, Cs, Ct);
            bool certain = true;
            (void)tri_positive(Cs, sx.r1, sx.r2, tx.n, tx.cn, certain);
            (void)tri_positive(Ct, tx.r1, tx.r2, sx.n, sx.cn, certain);
            sliver += !certain; accepted += want; ++n;
            if (got != want) { if (++bad < 10) std::printf("MISMATCH mode %d it %d want %d got %d\n", mode, it, (int)want, (int)got); }
        }
    std::printf("%lu cases, %lu accepted, %lu through the fallback, %lu mismatches -> %s\n", n, accepted, sliver, bad, bad ? "FAILED" : "identical");

    // ---- the float decision of round 4 (depths_positive32): whenever it reports `certain`, it must equal exact_depths ----
    // geometry as the match kernel sees it: unit rays / normals rounded to float, cn = C.n in double, B = C_t - C_s and
    // tolB as l3d_api.hip: pair_baseline computes them.  Modes: random; a ray within 1e-5 .. 1e-8 of the other plane
    // (da near zero); the baseline within 1e-5 .. 1e-9 of a plane (num near zero); camera centres of very different
    // magnitude (cancellation in cn); tiny baselines (tolB infinite: never certain); NaN.
    unsigned long n32 = 0, cert = 0, bad32 = 0, acc32 = 0;
    auto f3 = [](const double* v, float* o) { o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; };
    for (int mode = 0; mode < 6; ++mode)
        for (int it = 0; it < 400000; ++it) {
            SegX sx, tx; double Cs[3], Ct[3];
            unit(sx.r1); unit(sx.r2); unit(sx.n); unit(tx.r1); unit(tx.r2); unit(tx.n); unit(sx.rm); unit(tx.rm);
            const double scale = mode == 3 ? std::pow(10.0, 6 * U(rng)) : 10.0;
            for (int k = 0; k < 3; ++k) { Cs[k] = scale * U(rng); Ct[k] = (mode == 3 ? 10.0 : scale) * U(rng); }
            if (mode == 4) for (int k = 0; k < 3; ++k) Ct[k] = Cs[k] + std::pow(10.0, -2 - 4 * std::fabs(U(rng))) * U(rng);
            auto tilt = [&](double* v, const double* n, double eps) {   // make v . n = eps (v stays unit up to eps^2)
                const double d = v[0] * n[0] + v[1] * n[1] + v[2] * n[2];
                for (int k = 0; k < 3; ++k) v[k] -= (d - eps) * n[k];
                const double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                for (int k = 0; k < 3; ++k) v[k] /= l;
            };
            if (mode == 1) tilt((it & 1) ? sx.r1 : tx.r2, (it & 1) ? tx.n : sx.n, std::pow(10.0, -5 - 3 * std::fabs(U(rng))) * (U(rng) > 0 ? 1 : -1));
            if (mode == 2) {   // n_t (or n_s) nearly perpendicular to the baseline
                double B[3] = {Ct[0] - Cs[0], Ct[1] - Cs[1], Ct[2] - Cs[2]};
                const double l = std::sqrt(B[0] * B[0] + B[1] * B[1] + B[2] * B[2]);
                for (int k = 0; k < 3; ++k) B[k] /= l;
                tilt((it & 1) ? tx.n : sx.n, B, std::pow(10.0, -5 - 4 * std::fabs(U(rng))) * (U(rng) > 0 ? 1 : -1));
            }
            if (mode == 5 && (it % 5) == 0) tx.n[1] = NAN;
            sx.cn = Cs[0] * sx.n[0] + (Cs[1] * sx.n[1] + Cs[2] * sx.n[2]);
            tx.cn = Ct[0] * tx.n[0] + (Ct[1] * tx.n[1] + Ct[2] * tx.n[2]);
            PairResult res{};
            const bool want = exact_depths(sx, tx, Cs, Ct, res);
            SegD32 s32{}, t32{};
            f3(sx.r1, s32.r1); f3(sx.r2, s32.r2); f3(sx.n, s32.n); f3(tx.r1, t32.r1); f3(tx.r2, t32.r2); f3(tx.n, t32.n);
            // pair_baseline (l3d_api.hip), restated
            const double Bd[3] = {Ct[0] - Cs[0], Ct[1] - Cs[1], Ct[2] - Cs[2]};
            const double nb = std::sqrt(Bd[0] * Bd[0] + (Bd[1] * Bd[1] + Bd[2] * Bd[2]));
            double nc = 0; for (int k = 0; k < 3; ++k) nc += std::fabs(Cs[k]) + std::fabs(Ct[k]);
            const float B[3] = {(float)Bd[0], (float)Bd[1], (float)Bd[2]};
            const double tol = (double)kDepthTol32 * nb * (1.0 + 1e-6) + 1e-14 * nc;
            const float tolB = (nb >= 1e-3 && nb <= 1e30 && nc <= 1e30) ? std::nextafterf((float)tol, INFINITY) : INFINITY;
            bool certain = false;
            const bool got = depths_positive32(s32, t32, B, tolB, certain);
            ++n32; cert += certain; acc32 += want;
            if (certain && got != want) { if (++bad32 < 10) std::printf("MISMATCH32 mode %d it %d want %d got %d\n", mode, it, (int)want, (int)got); }
        }
    std::printf("float decision: %lu cases, %lu accepted, %lu certain, %lu mismatches among the certain -> %s\n", n32, acc32, cert, bad32,
                bad32 ? "FAILED" : "identical");
    return (bad || bad32) ? 1 : 0;
}
