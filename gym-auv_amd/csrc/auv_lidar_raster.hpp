// Building blocks of the work-efficient LiDAR in k_sense (auv_sense.hpp).
//
// Instead of testing every ray against every primitive
// (R*(C+S) exact tests per vessel), each primitive is "rasterised" onto the
// scan: a conservative interval of ray indices that can possibly hit it is
// derived from its bearing and angular width, and only those (primitive, ray)
// pairs run the exact fp64 intersection.  The pairs of a tile are laid out as
// one dense list (prefix sum of the interval lengths), so every lane of every
// wave does an exact test.  The nearest hit per ray is a 64-bit LDS atomic min
// on the bit pattern of t (t >= +0, so the order of the bits is the order of
// the values); a second sweep resolves the obstacle id: the lowest id among
// the primitives that attain the minimum.
//
// Exactness: the interval and range culls only decide which pairs are tried;
// every accept/reject and every reported number comes from the same exact
// test as the all-pairs kernel.  The culls use fp32 angles with an outward
// margin (0.02 ray spacings and at least 1e-4 rad) that is orders of
// magnitude above both the fp32 error (~2e-6 rad) and the
// deviation between the fp64 predicates and exact geometry (<= ~1e-8 rad), so
// no pair that the exact test would accept is ever skipped.
#pragma once
#include "auv_kernels.hpp"

namespace auv {

constexpr float kIvMargin = 0.02f;        // ray spacings added on each side, and at least ...
constexpr float kIvMarginRad = 1.0e-4f;   // ... this many radians (matters for very dense scans)
constexpr float kRangeSlack = 1.0001f;    // relative slack of the range cull

// atan2 for the candidate intervals only: odd minimax polynomial of the octant
// ratio, |error| < 2e-6 rad (fp32 evaluation included), no special cases: the
// callers never pass a vector shorter than 1e-3.  Far inside kIvMarginRad.
__device__ __forceinline__ float fast_atan2f(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float hi = fmaxf(ax, ay), lo = fminf(ax, ay);
    const float z = lo * __builtin_amdgcn_rcpf(hi);
    const float z2 = z * z;
    float p = -0.011719128f;  // fused steps: an estimate for the culls, not part of the exact arithmetic
    p = __builtin_fmaf(z2, p, 0.052647337f);
    p = __builtin_fmaf(z2, p, -0.116426475f);
    p = __builtin_fmaf(z2, p, 0.193540379f);
    p = __builtin_fmaf(z2, p, -0.332622826f);
    p = __builtin_fmaf(z2, p, 0.999977231f);
    float a = z * p;
    a = ay > ax ? 1.57079633f - a : a;
    a = x < 0.0f ? 3.14159265f - a : a;
    return y < 0.0f ? -a : a;
}

// square root for the candidate intervals only: the bare hardware instruction
// (1 ulp), without the scaling and the two correction steps that make sqrtf
// exact; its error is six orders of magnitude inside the interval margins
__device__ __forceinline__ float fast_sqrtf(float x) { return __builtin_amdgcn_sqrtf(x); }

// i / d for 0 <= i < 2^20 and 1 <= d < 2^16 through one fp32 multiply
// (inv_d = 1.0f / d): (i + 0.5) / d is at least 0.5/d away from an integer,
// far more than the rounding error of the product at these magnitudes.
__device__ __forceinline__ int fast_div(int i, float inv_d) { return (int)(((float)i + 0.5f) * inv_d); }

// fractional scan index of a world bearing: ray i sits at x = i, x in [-1, R-1)
__device__ __forceinline__ float scan_index(float bearing, float psi, float inv_delta) {
    float w = bearing - psi;
    const float two_pi = 6.2831853f, pi = 3.14159265f;
    w = __builtin_fmaf(-two_pi, floorf((w + pi) * (1.0f / two_pi)), w);
    return __builtin_fmaf(w + pi, inv_delta, -1.0f);
}

__device__ __forceinline__ int2 make_interval(float lo, float hi, int R, float margin) {
    const int ilo = (int)floorf(lo - margin);
    const int ihi = (int)ceilf(hi + margin);
    int cnt = ihi - ilo + 1;
    // ilo mod R; lo is never more than half a scan below -1, so for all but
    // the tiniest scans two conditional corrections replace the division
    int first;
    if (R < 8) {
        first = ilo % R;
        if (first < 0) first += R;
    } else {
        first = ilo < 0 ? ilo + R : ilo;
        first = first >= R ? first - R : first;
    }
    if (cnt >= R) {
        cnt = R;
        first = 0;
    }
    return make_int2(first, cnt);
}

// Ordering key of a hit parameter: the bit pattern of t as an unsigned integer
// is monotone for t >= +0; adding +0.0 first turns an accepted -0.0 (0 / -den)
// into +0.0, whose key is the smallest instead of the largest.
__device__ __forceinline__ unsigned long long t_key(double t) {
    return (unsigned long long)__double_as_longlong(t + 0.0);
}

// exact circle test; true and t when t >= 0 (the range is checked by the caller)
__device__ __forceinline__ bool circle_hit(double mx, double my, double cc, double dx, double dy, double& t) {
    const double b = mx * dx + my * dy;
    const double disc = b * b - cc;
    if (!(disc >= 0.0)) return false;
    const double sq = __builtin_sqrt(disc);
    double tt = b - sq;
    tt = tt < 0.0 ? b + sq : tt;
    t = tt;
    return tt >= 0.0;
}

// exact edge test
__device__ __forceinline__ bool edge_hit(double ax, double ay, double ex, double ey, double nt, double dx,
                                         double dy, double& t) {
    const double den = dx * ey - dy * ex;
    const double ns = ax * dy - ay * dx;
    const bool inseg = ((ns >= 0.0) & (ns <= den)) | ((ns <= 0.0) & (ns >= den));
    if (!(inseg && den != 0.0)) return false;
    const double tt = nt / den;
    t = tt;
    return tt >= 0.0;
}

}  // namespace auv
