// Device-side elementary functions of the simulation's arithmetic contract.
//
// Every result in the hot path must be reproducible bit for bit on any
// IEEE-754 binary64 machine, so that hit indices and collision flags can be
// compared exactly with a CPU restatement.  Therefore: no libm/ocml
// transcendental is called; sin/cos/atan2 are fixed sequences of unfused
// add/mul/div, and the translation unit is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace auv {

constexpr double kPi = 0x1.921fb54442d18p+1;
constexpr double kTwoPi = 0x1.921fb54442d18p+2;
constexpr double kInvTwoPi = 0x1.45f306dc9c883p-3;
constexpr double kPiO2 = 0x1.921fb54442d18p+0;
constexpr double kPiO4 = 0x1.921fb54442d18p-1;
constexpr double kPiO8 = 0x1.921fb54442d18p-2;
constexpr double kTwoOverPi = 0x1.45f306dc9c883p-1;
// pi/2 = HI + LO with 26 significant bits in HI: k*HI is exact for |k| < 2^27.
constexpr double kPiO2Hi = 0x1.921fb50000000p+0;
constexpr double kPiO2Lo = 0x1.110b4611a6263p-26;

struct SinCos {
    double s, c;
};

// Two-term Cody-Waite reduction to |r| <= pi/4, then Taylor polynomials in
// Horner form (degree 17 / 18; truncation < 1e-19).
__device__ __forceinline__ SinCos sincos_det(double x) {
    const double k = __builtin_rint(x * kTwoOverPi);
    const double r = (x - k * kPiO2Hi) - k * kPiO2Lo;
    const double r2 = r * r;
    double ps = 2.8114572543455206e-15;
    ps = -7.647163731819816e-13 + r2 * ps;
    ps = 1.6059043836821613e-10 + r2 * ps;
    ps = -2.505210838544172e-08 + r2 * ps;
    ps = 2.7557319223985893e-06 + r2 * ps;
    ps = -0.0001984126984126984 + r2 * ps;
    ps = 0.008333333333333333 + r2 * ps;
    ps = -0.16666666666666666 + r2 * ps;
    const double sr = r + (r * r2) * ps;
    double pc = -1.5619206968586225e-16;
    pc = 4.779477332387385e-14 + r2 * pc;
    pc = -1.1470745597729725e-11 + r2 * pc;
    pc = 2.08767569878681e-09 + r2 * pc;
    pc = -2.755731922398589e-07 + r2 * pc;
    pc = 2.48015873015873e-05 + r2 * pc;
    pc = -0.001388888888888889 + r2 * pc;
    pc = 0.041666666666666664 + r2 * pc;
    pc = -0.5 + r2 * pc;
    const double cr = 1.0 + r2 * pc;
    const int q = static_cast<int>(static_cast<long long>(k)) & 3;
    SinCos o;
    o.s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    o.c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
    return o;
}

// Angle to [-pi, pi).
__device__ __forceinline__ double wrap_angle(double a) {
    const double k = __builtin_floor((a + kPi) * kInvTwoPi);
    return a - k * kTwoPi;
}

// Four-quadrant arctangent; atan2_det(0, 0) = 0.  The octant ratio z in
// [0,1] is shifted to |w| <= tan(pi/16) around atan^-1{0, pi/8, pi/4} and
// summed with the alternating odd series through w^23.
__device__ __forceinline__ double atan2_det(double y, double x) {
    const double ax = __builtin_fabs(x);
    const double ay = __builtin_fabs(y);
    const bool zero = (ax == 0.0) && (ay == 0.0);
    const bool swap = ay > ax;
    const double num = swap ? ax : ay;
    double den = swap ? ay : ax;
    den = zero ? 1.0 : den;
    const double z = num / den;
    const bool hi = z > 0.6681786379192989;
    const bool mid = z > 0.198912367379658;
    const double c = hi ? 1.0 : (mid ? 0x1.a827999fcef32p-2 : 0.0);
    const double base = hi ? kPiO4 : (mid ? kPiO8 : 0.0);
    const double w = (z - c) / (1.0 + z * c);
    const double w2 = w * w;
    double p = -1.0 / 23.0;
    p = 1.0 / 21.0 + w2 * p;
    p = -1.0 / 19.0 + w2 * p;
    p = 1.0 / 17.0 + w2 * p;
    p = -1.0 / 15.0 + w2 * p;
    p = 1.0 / 13.0 + w2 * p;
    p = -1.0 / 11.0 + w2 * p;
    p = 1.0 / 9.0 + w2 * p;
    p = -1.0 / 7.0 + w2 * p;
    p = 1.0 / 5.0 + w2 * p;
    p = -1.0 / 3.0 + w2 * p;
    double a = base + (w + (w * w2) * p);
    a = swap ? (kPiO2 - a) : a;
    a = (x < 0.0) ? (kPi - a) : a;
    a = (y < 0.0) ? -a : a;
    return zero ? 0.0 : a;
}

// t / b, correctly rounded, for a divisor whose correctly rounded reciprocal
// y = RN(1/b) is known (b > 0, t >= 0): t*y refined twice with exact FMA
// residuals; the second refinement starts from a faithful quotient, which makes
// its result the rounded true quotient.  Quotients so small that a residual
// could underflow take the divider.
__device__ __forceinline__ double div_by_const(double t, double b, double y) {
    if (__builtin_expect(!(t >= 0x1p-900) && t != 0.0, 0)) return t / b;
    double q = t * y;
    double r = __builtin_fma(-q, b, t);
    q = __builtin_fma(r, y, q);
    r = __builtin_fma(-q, b, t);
    return __builtin_fma(r, y, q);
}

}  // namespace auv
