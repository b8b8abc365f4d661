// The reference's collision-risk index, as recovered from its dataset files
// (data/metadata_MovingObstaclesNoRules.csv against
// data/risk_MovingObstaclesNoRules.csv; record layout
// utils/explore_dataset.py:8,40-53): it reproduces all 23,680 per-obstacle
// records of those files (14,102 static, 9,578 moving) to 2e-15.
//
//   risk = max(0, 0.2 u_B(bearing) + 0.3 u_DCPA u_TCPA + 0.3 u_D(dist) + 0.2 u_A)
//
// dist is the reference's distance column: centre distance less the obstacle's
// size (a circle's radius, a ship's beam) less a clearance.  p is the obstacle
// position (dist along the world-frame direction of the obstacle's centre), v
// the unit direction of the relative course as the
// reference forms it, u_A = cos(a)|cos(a)| of the vessel's bearing a seen from
// a moving obstacle relative to that obstacle's heading (0 for a static one).
// Unfused fp64 in this operation order: bit-equal to the numpy and C
// restatements of the same index.
#pragma once
#include "auv_math.hpp"

namespace auv {

constexpr double kCriRad2Deg = 57.29577951308232;
constexpr double kCriD1 = 32.0, kCriD2 = 160.0;      // CPA distances: full risk / none
constexpr double kCriDa = 20.08, kCriDb = 45.18;     // range term: full risk / none
constexpr double kCriWb = 0.2, kCriWct = 0.3, kCriWd = 0.3, kCriWa = 0.2;

// quotients by constants: exact products for the powers of two, the correctly
// rounded quotient (div_by_const) for the others; bit-equal to a division
constexpr double kCriSpan = kCriDb - kCriDa;

__device__ __forceinline__ double cri_bearing_membership(double b) {
    const double bd = b * kCriRad2Deg;
    const double q1 = div_by_const(__builtin_fabs(90.0 - bd), 80.0, 1.0 / 80.0);
    const double q2 = div_by_const(__builtin_fabs(bd + 270.0), 225.0, 1.0 / 225.0);
    return bd < -45.0 ? q2 * q2 : (bd <= 10.0 ? 1.0 : (bd < 90.0 ? q1 * q1 : 0.0));
}

// The index is summed as two halves, clipped at 0:
// S = 0.3 u_DCPA u_TCPA + 0.3 u_D and T = 0.2 u_B + 0.2 u_A.
// along = p . v, cross = p x v for the obstacle position p and the unit
// relative-course direction v.
__device__ __forceinline__ double cri_cpa_range_half(double along, double cross, bool moving, double d) {
    const double dc = __builtin_fabs(cross);
    const double ln = __builtin_fabs(along);
    const bool near = dc <= kCriD1;
    const double lin = (kCriD2 - dc) * 0.0078125;   // / (D2 - D1)
    const double u_dc = near ? 1.0 : (lin > 0.0 ? lin : 0.0);
    const double r2 = kCriD2 * kCriD2 - dc * dc;
    const double t2 = __builtin_sqrt(r2 > 0.0 ? r2 : 0.0);
    const double r1 = kCriD1 * kCriD1 - dc * dc;
    const double t1 = near ? __builtin_sqrt(r1 > 0.0 ? r1 : 0.0) : kCriD1 - dc;
    const double den = t2 - t1;
    double q = (t2 - ln) / (den > 0.0 ? den : 1.0);
    q = q > 0.0 ? q : 0.0;
    q = q < 1.0 ? q : 1.0;
    const double u_ta = ln <= t1 ? 1.0 : q;
    const double o = 1.0 - ln * 0.03125;            // / D1
    const double u_tr = o > 0.0 ? o : 0.0;
    const double u_t = along < 0.0 ? u_ta : u_tr;
    const double ct = moving ? (kCriWct * u_dc) * u_t : 0.0;
    const double span = kCriDb - d;
    const double qd = span > 0.0 ? div_by_const(span, kCriSpan, 1.0 / kCriSpan) : 0.0;
    const double u_d = d <= kCriDa ? 1.0 : qd * qd;
    return ct + kCriWd * u_d;
}

__device__ __forceinline__ double cri_bearing_aspect_half(double b, double aspect) {
    return kCriWb * cri_bearing_membership(b) + kCriWa * aspect;
}

__device__ __forceinline__ double cri_sum(double s_half, double t_half) {
    const double total = s_half + t_half;
    return total > 0.0 ? total : 0.0;
}

__device__ __forceinline__ double cri_terms(double along, double cross, bool moving, double d, double b, double aspect) {
    return cri_sum(cri_cpa_range_half(along, cross, moving, d), cri_bearing_aspect_half(b, aspect));
}

// The relative motion of one obstacle the way the reference forms it: the
// vessel's velocity is its speed along its heading; an obstacle without a
// velocity is static (course along +x for a positive heading, -x otherwise);
// one with a velocity o is headed along it, and its relative velocity is o
// minus the vessel's, the latter mirrored about o when heading_ts - heading_OS
// leaves (-pi, pi].  With X = (cos psi, sin psi) x o that is: psi < 0 < o_y
// and X < 0, or o_y < 0 < psi and X > 0.
struct RefMotion {
    bool still;     // obstacle without a velocity
    double ss;      // the vessel's speed
    double wx, wy;  // a vector along the relative course ...
    double ww;      // ... and its squared length
    double rr;      // squared relative speed of a moving obstacle
};

__device__ __forceinline__ RefMotion reference_motion(double ovx, double ovy, double vox, double voy, double cpsi,
                                                      double spsi, double psi) {
    RefMotion m;
    m.ss = __builtin_sqrt(vox * vox + voy * voy);
    const double sx = m.ss * cpsi;
    const double sy = m.ss * spsi;
    m.still = ovx == 0.0 && ovy == 0.0;
    const double oo = ovx * ovx + ovy * ovy;
    const double X = cpsi * ovy - spsi * ovx;
    const bool mirror = (spsi < 0.0 && ovy > 0.0 && X < 0.0) || (spsi > 0.0 && ovy < 0.0 && X > 0.0);
    const double dot = sx * ovx + sy * ovy;
    const double f = (2.0 * dot) / (m.still ? 1.0 : oo);
    const double ex = mirror ? f * ovx - sx : sx;
    const double ey = mirror ? f * ovy - sy : sy;
    const double rx = ovx - ex;
    const double ry = ovy - ey;
    m.rr = rx * rx + ry * ry;
    m.wx = m.still ? (psi > 0.0 ? 1.0 : -1.0) : rx;
    m.wy = m.still ? 0.0 : ry;
    m.ww = m.still ? 1.0 : m.rr;
    return m;
}

// One obstacle of a simulator state: m = centre relative to the vessel, (ovx,
// ovy) the obstacle's velocity, (vox, voy) the vessel's, dist the reference's
// distance (centre distance less size less clearance), psi the vessel's heading
// with its cosine and sine.  The obstacle position is dist along m, so
// p . v = dist (m . w)/(|m||w|).  S half; -inf for an absent slot (!present),
// which makes the clipped sum 0.  Every present obstacle has a risk, whatever
// its distance.
__device__ __forceinline__ double risk_reference_s_of(const RefMotion& rm, double mx, double my, double dist, bool present) {
    const double mm = mx * mx + my * my;
    const double den = __builtin_sqrt(mm * rm.ww);
    const double dens = den > 0.0 ? den : 1.0;
    const bool moving = rm.still ? rm.ss > 0.0 : rm.rr > 0.0;
    const double along = dist * ((mx * rm.wx + my * rm.wy) / dens);
    const double cross = dist * ((mx * rm.wy - my * rm.wx) / dens);
    const double s_half = cri_cpa_range_half(along, cross, moving, dist);
    return present ? s_half : 0.0 - __builtin_inf();
}

__device__ __forceinline__ double risk_reference_s(double mx, double my, double ovx, double ovy, double vox, double voy,
                                                   double dist, double cpsi, double spsi, double psi, bool present) {
    return risk_reference_s_of(reference_motion(ovx, ovy, vox, voy, cpsi, spsi, psi), mx, my, dist, present);
}

// T half from the body bearing b = wrap(atan2(m) - psi); the aspect is
// -(m . o)|m . o| / (|m|^2 |o|^2) for an obstacle with a velocity.
__device__ __forceinline__ double risk_reference_t(double mx, double my, double ovx, double ovy, double b) {
    const double mo = mx * ovx + my * ovy;
    const double ad = (mx * mx + my * my) * (ovx * ovx + ovy * ovy);
    const bool has_ad = ad > 0.0;
    const double aspect = has_ad ? 0.0 - (mo * __builtin_fabs(mo)) / ad : 0.0;
    return cri_bearing_aspect_half(b, aspect);
}

__device__ __forceinline__ double risk_reference(double mx, double my, double ovx, double ovy, double vox, double voy,
                                                 double dist, double cpsi, double spsi, double psi, bool present) {
    const double b = wrap_angle(atan2_det(my, mx) - psi);
    return cri_sum(risk_reference_s(mx, my, ovx, ovy, vox, voy, dist, cpsi, spsi, psi, present),
                   risk_reference_t(mx, my, ovx, ovy, b));
}

// Dataset records -> risk: one thread per obstacle record of a metadata row.
struct RiskRowsArgs {
    long long n;
    const double* own_speed;
    const double* own_heading;
    const double* heading_ts;
    const double* dist;
    const double* rel_speed;
    const double* bearing;
    const double* course;
    double* out;
};

__global__ __launch_bounds__(256) void k_risk_rows(RiskRowsArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const double d = a.dist[i], b = a.bearing[i], ht = a.heading_ts[i], rs = a.rel_speed[i];
    const double bw = b + a.own_heading[i];
    const SinCos w = sincos_det(bw);
    const SinCos c = sincos_det(a.course[i]);
    const bool is_static = ht == 0.0 && rs == a.own_speed[i];  // utils/explore_dataset.py:46
    const double ca = sincos_det(bw - ht).c;                   // cos(a) = -cos(a - pi)
    const double aspect = is_static ? 0.0 : 0.0 - ca * __builtin_fabs(ca);
    const double px = d * w.c, py = d * w.s;
    a.out[i] = cri_terms(px * c.c + py * c.s, px * c.s - py * c.c, rs > 0.0, d, b, aspect);
}

}  // namespace auv
