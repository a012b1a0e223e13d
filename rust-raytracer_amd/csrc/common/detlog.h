// rtamd-ln-1: the natural logarithm used by ConstantMedium::hit (objects/medium.rs:38, `rng.gen::<f64>().ln()`).
//
// Rust's f64::ln is the platform libm's log; its last bit is not specified, and the device math library's differs from
// glibc's on some arguments.  The sampled free-flight distance decides whether a path scatters inside a medium, so host
// oracle and device kernel must agree to the bit: both evaluate THIS algorithm (the classic argument reduction
// x = 2^k (1 + f), sqrt(2)/2 < 1 + f < sqrt(2), s = f / (2 + f), log(1 + f) = 2 s + s R(s^2) with a degree-14 minimax R,
// as published for fdlibm's e_log.c; error < 1 ulp) with IEEE +, -, *, / only and contraction off.  The test oracle restates it
// independently, and tests/test_medium.py pins both against numpy's log to 1 ulp.
// Domain here: x in [0, 1) from gen::<f64>() (multiples of 2^-32, so never subnormal); x == 0 -> -inf; also correct for any
// finite normal x > 0.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RT_LN_HD __host__ __device__ __forceinline__
#else
#define RT_LN_HD inline
#endif

namespace rtamd {

RT_LN_HD double det_ln(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (x == 0.0) return -__builtin_inf();
    if (!(x > 0.0)) return __builtin_nan("");  // negative or NaN
    if (x == __builtin_inf()) return x;
    uint64_t b;
    memcpy(&b, &x, 8);
    int k = 0;
    if ((b >> 52) == 0) {  // subnormal: scale up by 2^54
        x *= 18014398509481984.0;
        memcpy(&b, &x, 8);
        k -= 54;
    }
    k += (int)(b >> 52) - 1023;
    // mantissa m in [1, 2); fold [sqrt(2), 2) down so that 1 + f lies in (sqrt(2)/2, sqrt(2))
    uint64_t mb = (b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &mb, 8);
    if (m > 1.41421356237309504880) {
        m = m * 0.5;
        k += 1;
    }
    const double f = m - 1.0;
    const double dk = (double)k;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

}  // namespace rtamd
