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
    return bad ? 1 : 0;
}
