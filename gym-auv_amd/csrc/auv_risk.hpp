// The reference's collision-risk index for static obstacles, as recovered
// from its dataset files (data/metadata_MovingObstaclesNoRules.csv against
// data/risk_MovingObstaclesNoRules.csv; record layout
// utils/explore_dataset.py:8,40-53): it reproduces all 14,102 static-obstacle
// records to 2e-15.  The records of moving obstacles are not reproduced; for
// them the index is an extrapolation.
//
//   risk = 0.2 u_B(bearing) + 0.3 u_DCPA u_TCPA + 0.3 u_D(dist)
//
// p is the obstacle position (hull distance along the world-frame direction
// of the obstacle), v the unit direction of the relative course; the
// reference takes 0 / -pi for that course of a static obstacle by the sign of
// the vessel's heading.  Unfused fp64 in this operation order: bit-equal to
// the numpy and C restatements of the same index.
#pragma once
#include "auv_math.hpp"

namespace auv {

constexpr double kCriRad2Deg = 57.29577951308232;
constexpr double kCriD1 = 32.0, kCriD2 = 160.0;      // CPA distances: full risk / none
constexpr double kCriDa = 20.08, kCriDb = 45.18;     // range term: full risk / none
constexpr double kCriWb = 0.2, kCriWct = 0.3, kCriWd = 0.3;

__device__ __forceinline__ double cri_bearing_membership(double b) {
    const double bd = b * kCriRad2Deg;
    const double q1 = (90.0 - bd) / 80.0;
    const double q2 = (bd + 270.0) / 225.0;
    return bd < -45.0 ? q2 * q2 : (bd <= 10.0 ? 1.0 : (bd < 90.0 ? q1 * q1 : 0.0));
}

__device__ __forceinline__ double cri_core(double px, double py, double vx, double vy, bool moving, double d, double b) {
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
    return (kCriWb * cri_bearing_membership(b) + ct) + kCriWd * u_d;
}

// One obstacle of a simulator state: m = centre relative to the vessel, (ovx,
// ovy) the obstacle's own velocity, rv its velocity relative to the vessel,
// dist the signed hull distance, psi the vessel's heading.
__device__ __forceinline__ double risk_reference(double mx, double my, double ovx, double ovy, double rvx, double rvy,
                                                 double dist, double psi, double range) {
    const double n = __builtin_sqrt(mx * mx + my * my);
    const bool has_n = n > 0.0;
    const double ns = has_n ? n : 1.0;
    const double ux = has_n ? mx / ns : 1.0;
    const double uy = has_n ? my / ns : 0.0;
    const double vv = rvx * rvx + rvy * rvy;
    const bool still = ovx == 0.0 && ovy == 0.0;
    const double sp = __builtin_sqrt(vv);
    const double sps = vv > 0.0 ? sp : 1.0;
    const double dvx = still ? (psi > 0.0 ? 1.0 : -1.0) : rvx / sps;
    const double dvy = still ? 0.0 : rvy / sps;
    const double b = wrap_angle(atan2_det(my, mx) - psi);
    const double r = cri_core(dist * ux, dist * uy, dvx, dvy, vv > 0.0, dist, b);
    return dist < range ? r : 0.0;
}

// Dataset records -> risk: one thread per record.
struct RiskRowsArgs {
    long long n;
    const double* own_heading;
    const double* dist;
    const double* rel_speed;
    const double* bearing;
    const double* course;
    double* out;
};

__global__ __launch_bounds__(256) void k_risk_rows(RiskRowsArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const double d = a.dist[i], b = a.bearing[i];
    const SinCos w = sincos_det(b + a.own_heading[i]);
    const SinCos c = sincos_det(a.course[i]);
    a.out[i] = cri_core(d * w.c, d * w.s, c.c, c.s, a.rel_speed[i] > 0.0, d, b);
}

}  // namespace auv
