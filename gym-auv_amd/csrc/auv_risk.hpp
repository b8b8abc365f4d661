// The reference's collision-risk index, as recovered from its dataset files
// (data/metadata_MovingObstaclesNoRules.csv against
// data/risk_MovingObstaclesNoRules.csv; record layout
// utils/explore_dataset.py:8,40-53): it reproduces all 23,680 per-obstacle
// records of those files (14,102 static, 9,578 moving) to 2e-15.
//
//   risk = max(0, 0.2 u_B(bearing) + 0.3 u_DCPA u_TCPA + 0.3 u_D(dist) + 0.2 u_A)
//
// p is the obstacle position (hull distance along the world-frame direction
// of the obstacle), v the unit direction of the relative course as the
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

__device__ __forceinline__ double cri_bearing_membership(double b) {
    const double bd = b * kCriRad2Deg;
    const double q1 = (90.0 - bd) / 80.0;
    const double q2 = (bd + 270.0) / 225.0;
    return bd < -45.0 ? q2 * q2 : (bd <= 10.0 ? 1.0 : (bd < 90.0 ? q1 * q1 : 0.0));
}

__device__ __forceinline__ double cri_core(double px, double py, double vx, double vy, bool moving, double d, double b,
                                           double aspect) {
    const double along = px * vx + py * vy;
    const double cross = px * vy - py * vx;
    const double dc = __builtin_fabs(cross);
    const double ln = __builtin_fabs(along);
    const bool near = dc <= kCriD1;
    const double lin = (kCriD2 - dc) / (kCriD2 - kCriD1);
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
    const double o = 1.0 - ln / kCriD1;
    const double u_tr = o > 0.0 ? o : 0.0;
    const double u_t = along < 0.0 ? u_ta : u_tr;
    const double ct = moving ? (kCriWct * u_dc) * u_t : 0.0;
    double qd = (kCriDb - d) / (kCriDb - kCriDa);
    qd = qd > 0.0 ? qd : 0.0;
    const double u_d = d <= kCriDa ? 1.0 : qd * qd;
    const double total = ((kCriWb * cri_bearing_membership(b) + ct) + kCriWd * u_d) + kCriWa * aspect;
    return total > 0.0 ? total : 0.0;
}

// The relative motion of one obstacle the way the reference forms it: the
// vessel's velocity is its speed along its heading; an obstacle without a
// velocity is static (course 0 for a positive heading, -pi otherwise, relative
// speed = the vessel's speed); one with a velocity is headed along it, and its
// relative velocity is its own minus the vessel's, the latter mirrored about
// the obstacle's heading when heading_ts - heading_OS leaves (-pi, pi].
struct RefMotion {
    bool still;      // obstacle without a velocity
    double hx, hy;   // unit heading of a moving obstacle
    double heading;  // heading_ts: atan2 of its velocity, 0 when still
    double rx, ry;   // relative velocity (moving obstacle)
    double speed;    // relative speed
    double course;   // logged relative course, (-pi, pi]
};

__device__ __forceinline__ RefMotion reference_motion(double ovx, double ovy, double vox, double voy, double psi) {
    RefMotion m;
    const SinCos sc = sincos_det(psi);
    const double ss = __builtin_sqrt(vox * vox + voy * voy);
    const double sx = ss * sc.c;
    const double sy = ss * sc.s;
    m.still = ovx == 0.0 && ovy == 0.0;
    const double so = __builtin_sqrt(ovx * ovx + ovy * ovy);
    const double sos = m.still ? 1.0 : so;
    m.hx = ovx / sos;
    m.hy = ovy / sos;
    m.heading = atan2_det(ovy, ovx);
    const bool mirror = __builtin_fabs(m.heading - psi) > kPi;
    const double dot = sx * m.hx + sy * m.hy;
    const double ex = mirror ? (2.0 * dot) * m.hx - sx : sx;
    const double ey = mirror ? (2.0 * dot) * m.hy - sy : sy;
    m.rx = ovx - ex;
    m.ry = ovy - ey;
    const double rn = __builtin_sqrt(m.rx * m.rx + m.ry * m.ry);
    m.speed = m.still ? ss : rn;
    m.course = m.still ? (psi > 0.0 ? 0.0 : 0.0 - kPi) : atan2_det(m.ry, m.rx);
    return m;
}

// One obstacle of a simulator state: m = centre relative to the vessel, (ovx,
// ovy) the obstacle's velocity, (vox, voy) the vessel's, dist the signed hull
// distance, psi the vessel's heading.
__device__ __forceinline__ double risk_reference(double mx, double my, double ovx, double ovy, double vox, double voy,
                                                 double dist, double psi, double range) {
    const RefMotion rm = reference_motion(ovx, ovy, vox, voy, psi);
    const double n = __builtin_sqrt(mx * mx + my * my);
    const bool has_n = n > 0.0;
    const double ns = has_n ? n : 1.0;
    const double ux = has_n ? mx / ns : 1.0;
    const double uy = has_n ? my / ns : 0.0;
    const double rns = rm.speed > 0.0 ? rm.speed : 1.0;
    const double dvx = rm.still ? (psi > 0.0 ? 1.0 : -1.0) : rm.rx / rns;
    const double dvy = rm.still ? 0.0 : rm.ry / rns;
    const double ca = 0.0 - (ux * rm.hx + uy * rm.hy);
    const double aspect = rm.still ? 0.0 : ca * __builtin_fabs(ca);
    const double b = wrap_angle(atan2_det(my, mx) - psi);
    const double r = cri_core(dist * ux, dist * uy, dvx, dvy, rm.speed > 0.0, dist, b, aspect);
    return dist < range ? r : 0.0;
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
    a.out[i] = cri_core(d * w.c, d * w.s, c.c, c.s, rs > 0.0, d, b, aspect);
}

}  // namespace auv
