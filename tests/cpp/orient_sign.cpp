// CPU pin of l3d_dev.h: orientation_ok_fast (checkMatchOrientation's decision, line3D.cc:811-858 / view.cc:466-484, without
// the square roots and divisions of unprojectSegment) against orientation_ok (the reference's arithmetic): random
// geometry, directions forced to the neighbourhood of both thresholds (where the fallback must decide), zero-length and
// near-zero-length segments, NaN.
#include <cmath>
#include <cstdio>
#include <random>

#include "../../line3dpp_amd/csrc/l3d_dev.h"

using namespace l3d;

int main() {
    // the thresholds as l3d_api.hip finds them: largest dp with acos(dp) > PI/32, smallest with acos(dp) < 31 PI/32
    const double a1 = (double)kPi_1_32, a2 = (double)kPi_31_32;
    double x = -1.0, y = 1.0;
    for (int it = 0; it < 200 && std::nextafter(x, y) < y; ++it) { const double m = 0.5 * (x + y); if (std::acos(m) > a1) x = m; else y = m; }
    const double hi = x;
    x = -1.0; y = 1.0;
    for (int it = 0; it < 200 && std::nextafter(x, y) < y; ++it) { const double m = 0.5 * (x + y); if (std::acos(m) < a2) y = m; else x = m; }
    const double lo = y;
    const OrientThr thr{lo, hi};
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    auto unit = [&](double* v) { double a = U(rng), b = U(rng), c = U(rng), l = std::sqrt(a * a + b * b + c * c) + 1e-30; v[0] = a / l; v[1] = b / l; v[2] = c / l; };
    unsigned long n = 0, kept = 0, bad = 0, exact_path = 0;
    for (int mode = 0; mode < 5; ++mode)
        for (int it = 0; it < 500000; ++it) {
            SegX sx; double C[3];
            unit(sx.r1); unit(sx.r2); unit(sx.rm); unit(sx.n); sx.cn = 0;
            for (int k = 0; k < 3; ++k) C[k] = 20 * U(rng);
            float d1 = (float)(30 * std::fabs(U(rng)) + 0.1), d2 = (float)(30 * std::fabs(U(rng)) + 0.1);
            if (mode == 1 || mode == 2) {   // mid-point ray at the threshold angle from the segment direction (+- a few ulp)
                const Seg3 s = unproject(C, sx.r1, sx.r2, d1, d2);
                double p[3]; unit(p);
                // orthonormalise p against dir
                const double dd = p[0] * s.dir.x + p[1] * s.dir.y + p[2] * s.dir.z;
                double q[3] = {p[0] - dd * s.dir.x, p[1] - dd * s.dir.y, p[2] - dd * s.dir.z};
                const double ql = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
                double cth = mode == 1 ? hi : lo;
                cth += (double)((int)(rng() % 41) - 20) * 1.1e-16 * (1 + (rng() % 1000 == 0 ? 1e4 : 0));
                const double sth = std::sqrt(std::fmax(0.0, 1 - cth * cth));
                sx.rm[0] = cth * s.dir.x + sth * q[0] / ql; sx.rm[1] = cth * s.dir.y + sth * q[1] / ql; sx.rm[2] = cth * s.dir.z + sth * q[2] / ql;
            }
            if (mode == 3) { for (int k = 0; k < 3; ++k) sx.r2[k] = sx.r1[k]; d2 = d1 * (1.0f + (it % 3 == 0 ? 0.0f : 1e-7f * (float)U(rng))); if (it % 5 == 0) { d1 = d2 = 1e-13f; } }
            if (mode == 4 && it % 9 == 0) d1 = NAN;
            const bool want = orientation_ok(C, sx, d1, d2, thr);
            const bool got = orientation_ok_fast(C, sx, d1, d2, thr);
            kept += want; ++n;
            if (got != want && ++bad < 10) std::printf("MISMATCH mode %d it %d want %d got %d\n", mode, it, (int)want, (int)got);
        }
    (void)exact_path;
    std::printf("lo %.17g hi %.17g: %lu cases, %lu kept, %lu mismatches -> %s\n", lo, hi, n, kept, bad, bad ? "FAILED" : "identical");
    return bad ? 1 : 0;
}
