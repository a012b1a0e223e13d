// rtamd-sin-1: the sine of the Perlin marble texture (book 2's noise_texture: 0.5 (1 + sin(scale p.z + 10 turb(p)))), a book-2
// extension the reference has no code for.
//
// libm's sin is not correctly rounded and its last bit differs between glibc and the device math library; the value becomes a
// colour, so host oracle and device kernel must agree to the bit: both evaluate THIS algorithm -- Cody-Waite reduction
// x = n pi/2 + (y0 + y1) with the two-step constants published for fdlibm's e_rem_pio2.c (always two steps: good to 118 bits, valid
// for |x| < 2^19 pi/2) and the degree-13 / degree-14 kernels of k_sin.c / k_cos.c in their plain forms -- with IEEE + - * only and
// contraction off.  Error < 1 ulp on the range the texture uses (max |error| 1.1e-16 against libm on 3e5 arguments,
// tests/test_book2.py).  |x| beyond the range (or NaN) -> 0.  The test oracle restates it independently.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_SIN_HD __host__ __device__ __forceinline__
#else
#define RT_SIN_HD inline
#endif

namespace rtamd {

RT_SIN_HD double det_sin(double x) {
    const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_2 = 6.07710050630396597660e-11,
                 pio2_2t = 2.02226624879595063154e-21;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double t = x < 0. ? -x : x;
    if (!(t < 823549.0)) return 0.0;
    const int n = (int)(t * invpio2 + 0.5);
    const double fn = (double)n;
    const double r1 = t - fn * pio2_1;
    const double w2 = fn * pio2_2;
    const double r2 = r1 - w2;
    const double w = fn * pio2_2t - ((r1 - r2) - w2);
    const double y0 = r2 - w, y1 = (r2 - y0) - w;
    const double z = y0 * y0;
    double res;
    if ((n & 1) == 0) {
        const double v = z * y0;
        const double rr = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
        res = y0 - ((z * (0.5 * y1 - v * rr) - y1) - v * S1);
    } else {
        const double rr = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
        res = 1.0 - (0.5 * z - (z * rr - y0 * y1));
    }
    if (n & 2) res = -res;
    return (x < 0.) ? -res : res;
}

}  // namespace rtamd
