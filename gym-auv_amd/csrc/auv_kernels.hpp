// HIP kernels of the simulation step for gfx950 (64-wide wavefronts, no MFMA:
// nothing on this path is a dense contraction).
//
//   k_dynamics   one thread per vessel: RKF45 step of the 3-DOF model on SoA
//                fp64 state (coalesced 8-B lanes); the same launch advances
//                the constant-velocity obstacles.  112 B / vessel; latency-sized.
//   k_encounter  one thread per (env, obstacle): signed hull distance,
//                encounter metadata, risk; then one thread per env: collision
//                flag, nearest obstacle, navigation features, done flags.
//   k_lidar      one workgroup per tile of envs: obstacle primitives are made
//                vessel-relative once and staged in LDS, then one thread per
//                (vessel, ray) walks them and keeps the nearest hit.  fp64
//                VALU-bound; 16 B written per ray.
//
// All arithmetic follows the contract in auv_math.hpp (unfused fp64, fixed
// operation order) so that results equal a CPU restatement bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "auv_math.hpp"
#include "auv_risk.hpp"

namespace auv {

enum ModelIdx {
    MP_MINV11 = 0, MP_MINV22, MP_MINV23, MP_MINV32, MP_MINV33,
    MP_MU, MP_MXG, MP_RESERVED,
    MP_XU, MP_XUU, MP_YV, MP_YVV, MP_YR, MP_NV, MP_NR, MP_NRR,
    MP_COUNT
};

struct Dims {
    int V, R, C, P, NV, O;
};

// ---------------------------------------------------------------------------
// dynamics
// ---------------------------------------------------------------------------
struct DynArgs {
    Dims d;
    double* state;         // [6,V]
    const double* action;  // [2,V]
    int32_t* t_step;       // [V]
    double* circles;       // [V,C,5]
    double* poly_pose;     // [V,P,6]
    double dt, thrust_max, moment_max;
    double mp[MP_COUNT];
    int advance;  // 1: also advance the obstacles here (0 when k_sense does it)
};

struct Vec6 {
    double x, y, psi, u, v, r;
};

__device__ __forceinline__ Vec6 vessel_rhs(const Vec6& s, double fx, double mz, const double* mp) {
    const SinCos sc = sincos_det(s.psi);
    Vec6 o;
    o.x = sc.c * s.u - sc.s * s.v;
    o.y = sc.s * s.u + sc.c * s.v;
    o.psi = s.r;
    // Linearised manoeuvring model: the velocity-dependent matrix N(nu) holds the
    // linear damping plus the rigid-body terms m*u (sway) and m*x_g*u (yaw) that
    // multiply the yaw rate; there is no v*r term in surge.
    const double n1 = (mp[MP_XU] + mp[MP_XUU] * __builtin_fabs(s.u)) * s.u;
    const double n2 = (mp[MP_YV] + mp[MP_YVV] * __builtin_fabs(s.v)) * s.v + (mp[MP_MU] * s.u + mp[MP_YR]) * s.r;
    const double n3 = mp[MP_NV] * s.v + ((mp[MP_MXG] * s.u + mp[MP_NR]) + mp[MP_NRR] * __builtin_fabs(s.r)) * s.r;
    const double r1 = fx - n1;
    const double r2 = 0.0 - n2;
    const double r3 = mz - n3;
    o.u = mp[MP_MINV11] * r1;
    o.v = mp[MP_MINV22] * r2 + mp[MP_MINV23] * r3;
    o.r = mp[MP_MINV32] * r2 + mp[MP_MINV33] * r3;
    return o;
}

// y + h * acc, component-wise
__device__ __forceinline__ Vec6 axpy6(const Vec6& y, double h, const Vec6& a) {
    Vec6 o;
    o.x = y.x + h * a.x;
    o.y = y.y + h * a.y;
    o.psi = y.psi + h * a.psi;
    o.u = y.u + h * a.u;
    o.v = y.v + h * a.v;
    o.r = y.r + h * a.r;
    return o;
}

#define AUV_EACH6(M) M(x) M(y) M(psi) M(u) M(v) M(r)

// One Fehlberg 4(5) step of size h (5th-order solution propagated), heading
// wrapped afterwards; actions are clipped to [0,1] x [-1,1] first.
__device__ __forceinline__ Vec6 rkf45_step(const Vec6& s, double a0, double a1, double h, double thrust_max,
                                           double moment_max, const double* mp) {
    a0 = a0 < 0.0 ? 0.0 : a0;
    a0 = a0 > 1.0 ? 1.0 : a0;
    a1 = a1 < -1.0 ? -1.0 : a1;
    a1 = a1 > 1.0 ? 1.0 : a1;
    const double fx = a0 * thrust_max;
    const double mz = a1 * moment_max;
    const Vec6 k0 = vessel_rhs(s, fx, mz, mp);
    Vec6 acc;
#define M(f) acc.f = (1.0 / 4.0) * k0.f;
    AUV_EACH6(M)
#undef M
    const Vec6 k1 = vessel_rhs(axpy6(s, h, acc), fx, mz, mp);
#define M(f) acc.f = (3.0 / 32.0) * k0.f + (9.0 / 32.0) * k1.f;
    AUV_EACH6(M)
#undef M
    const Vec6 k2 = vessel_rhs(axpy6(s, h, acc), fx, mz, mp);
#define M(f) acc.f = ((1932.0 / 2197.0) * k0.f + (-7200.0 / 2197.0) * k1.f) + (7296.0 / 2197.0) * k2.f;
    AUV_EACH6(M)
#undef M
    const Vec6 k3 = vessel_rhs(axpy6(s, h, acc), fx, mz, mp);
#define M(f) \
    acc.f = (((439.0 / 216.0) * k0.f + (-8.0) * k1.f) + (3680.0 / 513.0) * k2.f) + (-845.0 / 4104.0) * k3.f;
    AUV_EACH6(M)
#undef M
    const Vec6 k4 = vessel_rhs(axpy6(s, h, acc), fx, mz, mp);
#define M(f)                                                                                   \
    acc.f = ((((-8.0 / 27.0) * k0.f + 2.0 * k1.f) + (-3544.0 / 2565.0) * k2.f) +               \
             (1859.0 / 4104.0) * k3.f) +                                                       \
            (-11.0 / 40.0) * k4.f;
    AUV_EACH6(M)
#undef M
    const Vec6 k5 = vessel_rhs(axpy6(s, h, acc), fx, mz, mp);
#define M(f)                                                                                   \
    acc.f = ((((16.0 / 135.0) * k0.f + (6656.0 / 12825.0) * k2.f) + (28561.0 / 56430.0) * k3.f) + \
             (-9.0 / 50.0) * k4.f) +                                                           \
            (2.0 / 55.0) * k5.f;
    AUV_EACH6(M)
#undef M
    Vec6 o = axpy6(s, h, acc);
    o.psi = wrap_angle(o.psi);
    return o;
}

__device__ __forceinline__ Vec6 load_state(const double* state, int V, int i) {
    Vec6 s;
    s.x = state[0 * (size_t)V + i];
    s.y = state[1 * (size_t)V + i];
    s.psi = state[2 * (size_t)V + i];
    s.u = state[3 * (size_t)V + i];
    s.v = state[4 * (size_t)V + i];
    s.r = state[5 * (size_t)V + i];
    return s;
}

__device__ __forceinline__ void store_state(double* state, int V, int i, const Vec6& o) {
    state[0 * (size_t)V + i] = o.x;
    state[1 * (size_t)V + i] = o.y;
    state[2 * (size_t)V + i] = o.psi;
    state[3 * (size_t)V + i] = o.u;
    state[4 * (size_t)V + i] = o.v;
    state[5 * (size_t)V + i] = o.r;
}

__global__ __launch_bounds__(256) void k_dynamics(DynArgs a) {
    const int V = a.d.V;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V) {
        const Vec6 s = load_state(a.state, V, i);
        const Vec6 o = rkf45_step(s, a.action[i], a.action[(size_t)V + i], a.dt, a.thrust_max, a.moment_max, a.mp);
        store_state(a.state, V, i, o);
        a.t_step[i] = a.t_step[i] + 1;
    }
    // constant-velocity obstacle advance: slot i of the flattened
    // [V*C circles | V*P polygons] list
    const long long n_c = (long long)V * a.d.C;
    const long long n_p = (long long)V * a.d.P;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = i; a.advance && j < n_c + n_p; j += stride) {
        if (j < n_c) {
            double* c = a.circles + j * 5;
            c[0] = c[0] + c[3] * a.dt;
            c[1] = c[1] + c[4] * a.dt;
        } else {
            double* p = a.poly_pose + (j - n_c) * 6;
            p[0] = p[0] + p[4] * a.dt;
            p[1] = p[1] + p[5] * a.dt;
        }
    }
}

// masked reset: vessel state, step counter, and the env's obstacles back to
// the poses they were uploaded with
// A path is a polyline of up to KS segments per env, one record of kPathRec
// doubles each: x0, y0, angle, cos angle, sin angle, length, arc length before
// the segment, 0.  A vessel is ON one segment (carried per env): when its
// projection passes the end of that segment it moves on to the next, never
// back.  Along-track position = arc length before the segment + projection on
// it; cross-track and heading errors are those of the current segment; the
// look-ahead point lies look_ahead further along the path, on the segment that
// holds that arc length (the last one extended).  One segment reproduces the
// straight-path arithmetic exactly (0 + x and x - 0 are exact).
constexpr int kPathRec = 8;
struct PathNav {
    double along, cte, la, he, total;
};
struct PathRecord {
    double x0, y0, angle, ca, sa, len, before;
};
__device__ __forceinline__ PathRecord load_path_record(const double* tab, int seg) {
    const double* pt = tab + (size_t)seg * kPathRec;
    return PathRecord{pt[0], pt[1], pt[2], pt[3], pt[4], pt[5], pt[6]};
}
// `cur` = the record of segment `seg` and `last` = the record of the last segment in use, both loaded by the
// caller (k_vessel asks for them before its integration, whose length hides their latency)
__device__ __forceinline__ PathNav path_navigation(const double* tab, int nseg, int& seg, PathRecord cur,
                                                   const PathRecord& last, double px, double py, double psi,
                                                   double look_ahead) {
    double rx = px - cur.x0, ry = py - cur.y0;
    double loc = rx * cur.ca + ry * cur.sa;
    while (seg + 1 < nseg && loc > cur.len) {
        ++seg;
        cur = load_path_record(tab, seg);
        rx = px - cur.x0;
        ry = py - cur.y0;
        loc = rx * cur.ca + ry * cur.sa;
    }
    PathNav n;
    n.along = cur.before + loc;
    n.cte = ry * cur.ca - rx * cur.sa;
    const double st = n.along + look_ahead;
    PathRecord pl = cur;
    for (int j = seg; j + 1 < nseg && st > pl.before + pl.len; ++j) pl = load_path_record(tab, j + 1);
    const double tx = pl.x0 + (st - pl.before) * pl.ca;
    const double ty = pl.y0 + (st - pl.before) * pl.sa;
    n.la = wrap_angle(atan2_det(ty - py, tx - px) - psi);
    n.he = wrap_angle(cur.angle - psi);
    n.total = last.before + last.len;
    return n;
}

struct ResetArgs {
    Dims d;
    double* state;
    int32_t* t_step;
    int32_t* path_seg;    // [V] back to the first segment
    const double* init;   // [6,V]
    const uint8_t* mask;  // [V] or null = all
    double* circles;
    const double* circles0;
    double* poly_pose;
    const double* poly_pose0;
};

__global__ __launch_bounds__(256) void k_reset(ResetArgs a) {
    const int V = a.d.V;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    if (a.mask != nullptr && a.mask[i] == 0) return;
    for (int f = 0; f < 6; ++f) a.state[f * (size_t)V + i] = a.init[f * (size_t)V + i];
    a.t_step[i] = 0;
    a.path_seg[i] = 0;
    const size_t c0 = (size_t)i * a.d.C * 5;
    for (int k = 0; k < a.d.C * 5; ++k) a.circles[c0 + k] = a.circles0[c0 + k];
    const size_t p0 = (size_t)i * a.d.P * 6;
    for (int k = 0; k < a.d.P * 6; ++k) a.poly_pose[p0 + k] = a.poly_pose0[p0 + k];
}

// ---------------------------------------------------------------------------
// encounter: distances, metadata, risk, per-env flags and navigation
// ---------------------------------------------------------------------------
struct EncArgs {
    Dims d;
    int EB;  // envs per block
    const double* state;
    const int32_t* t_step;
    const double* circles;
    const double* poly_pose;
    const double* poly_verts;  // [V,P,NV,2]
    const int32_t* poly_nv;    // [V,P]
    const double* poly_heading;
    const double* path;  // [V,KS,kPathRec]
    const int32_t* path_nseg;  // [V]
    int32_t* path_seg;         // [V]
    int KS;
    double* obst_dist;   // [V,O]
    double* risk;        // [V,O]
    double* meta;        // [V,O,5]
    double* own;         // [V,2]
    uint8_t* collision;
    uint8_t* reached_goal;
    uint8_t* truncated;
    uint8_t* done;
    double* min_dist;
    int32_t* min_idx;
    double* along;
    double* reward;
    float* navigation;  // [V,1,6]
    double sensor_range, vessel_radius, look_ahead, goal_radius;
    double contact_dist;
    double w_d, w_c, t_c;
    double rw_progress, rw_collision, rw_goal, rw_cte;
    int max_steps;
    int is_step;  // 1: called from a step (reward from progress), 0: after reset
    int risk_model;  // 0: the closest-point-of-approach index, 1: the reference's index (auv_risk.hpp)
};

__device__ __forceinline__ double poly_signed_distance(const double* pose, const double* verts, int nv,
                                                       double px, double py) {
    const double X = pose[0], Y = pose[1], ch = pose[2], sh = pose[3];
    double dmin = __builtin_inf();
    bool inside = false;
    const double w0x = X + (ch * verts[0] - sh * verts[1]);
    const double w0y = Y + (sh * verts[0] + ch * verts[1]);
    double Ax = w0x, Ay = w0y;
    for (int k = 0; k < nv; ++k) {
        double Bx, By;
        if (k + 1 < nv) {
            const double bx = verts[2 * (k + 1)], by = verts[2 * (k + 1) + 1];
            Bx = X + (ch * bx - sh * by);
            By = Y + (sh * bx + ch * by);
        } else {
            Bx = w0x;
            By = w0y;
        }
        const double ax = Ax - px;
        const double ay = Ay - py;
        const double by = By - py;
        const double ex = Bx - Ax;
        const double ey = By - Ay;
        const double ee = ex * ex + ey * ey;
        double tp = 0.0;
        if (ee > 0.0) {
            tp = (0.0 - (ax * ex + ay * ey)) / ee;
            tp = tp < 0.0 ? 0.0 : tp;
            tp = tp > 1.0 ? 1.0 : tp;
        }
        const double qx = ax + tp * ex;
        const double qy = ay + tp * ey;
        const double dd = __builtin_sqrt(qx * qx + qy * qy);
        dmin = dd < dmin ? dd : dmin;
        const double nt = ax * ey - ay * ex;
        const bool up = (ay <= 0.0) && (by > 0.0) && (nt > 0.0);
        const bool dn = (by <= 0.0) && (ay > 0.0) && (nt < 0.0);
        inside ^= (up || dn);
        Ax = Bx;
        Ay = By;
    }
    return inside ? -dmin : dmin;
}

// Beam of a ship: max - min of the body-frame y of its nv >= 1 vertices.
__device__ __forceinline__ double body_beam(const double* verts, int nv) {
    double hi = -__builtin_inf(), lo = __builtin_inf();
    for (int k = 0; k < nv; ++k) {
        const double y = verts[2 * k + 1];
        hi = y > hi ? y : hi;
        lo = y < lo ? y : lo;
    }
    return hi - lo;
}

__global__ __launch_bounds__(256) void k_encounter(EncArgs a) {
    extern __shared__ double lds_dist[];  // [EB*O]
    const int V = a.d.V, C = a.d.C, P = a.d.P, NV = a.d.NV, O = a.d.O;
    const int e0 = blockIdx.x * a.EB;
    const int ne = min(a.EB, V - e0);
    for (int i = threadIdx.x; i < ne * O; i += blockDim.x) {
        const int el = i / O;
        const int o = i - el * O;
        const int e = e0 + el;
        const double px = a.state[e];
        const double py = a.state[(size_t)V + e];
        const double psi = a.state[2 * (size_t)V + e];
        const double u = a.state[3 * (size_t)V + e];
        const double v = a.state[4 * (size_t)V + e];
        const SinCos sc = sincos_det(psi);
        const double vox = sc.c * u - sc.s * v;
        const double voy = sc.s * u + sc.c * v;
        double cx, cy, vx, vy, hd, dist;
        double beam = -1.0;  // ships only: extent of the body-frame vertices across the heading
        if (o < C) {
            const double* c = a.circles + ((size_t)e * C + o) * 5;
            cx = c[0];
            cy = c[1];
            const double rad = c[2];
            vx = c[3];
            vy = c[4];
            hd = 0.0;
            const double mx = cx - px;
            const double my = cy - py;
            const double dd = (__builtin_sqrt(mx * mx + my * my) - rad) - a.vessel_radius;
            dist = rad > 0.0 ? dd : __builtin_inf();
        } else {
            const int q = o - C;
            const double* p = a.poly_pose + ((size_t)e * P + q) * 6;
            cx = p[0];
            cy = p[1];
            vx = p[4];
            vy = p[5];
            hd = a.poly_heading[(size_t)e * P + q];
            const int nv = a.poly_nv[(size_t)e * P + q];
            if (nv > 0) {
                const double* verts = a.poly_verts + ((size_t)e * P + q) * NV * 2;
                dist = poly_signed_distance(p, verts, nv, px, py) - a.vessel_radius;
                beam = body_beam(verts, nv);
            } else {
                dist = __builtin_inf();
            }
        }
        const double mx = cx - px;
        const double my = cy - py;
        const double rvx = vx - vox;
        const double rvy = vy - voy;
        const double vv = rvx * rvx + rvy * rvy;
        const size_t eo = (size_t)e * O + o;
        a.obst_dist[eo] = dist;
        lds_dist[i] = dist;
        double* m = a.meta + eo * 5;
        // the reference logs centre distance less size less clearance, for ships too
        const double dref = (a.risk_model == 1 && beam >= 0.0)
                                ? (__builtin_sqrt(mx * mx + my * my) - beam) - a.vessel_radius : dist;
        m[1] = dref;
        m[3] = wrap_angle(atan2_det(my, mx) - psi);
        if (a.risk_model == 1) {
            // speed, heading and course columns follow the reference's conventions
            const RefMotion rm = reference_motion(vx, vy, vox, voy, sc.c, sc.s, psi);
            m[0] = rm.still ? 0.0 : atan2_det(vy, vx);
            m[2] = rm.still ? rm.ss : __builtin_sqrt(rm.rr);
            m[4] = rm.still ? (psi > 0.0 ? 0.0 : 0.0 - kPi) : atan2_det(rm.wy, rm.wx);
        } else {
            m[0] = hd;
            m[2] = __builtin_sqrt(vv);
            m[4] = wrap_angle(atan2_det(rvy, rvx) - psi);
        }
        // closest-point-of-approach index (AUV_RISK_CPA)
        const double R = a.sensor_range;
        const double d = dist < 0.0 ? 0.0 : dist;
        const bool inrange = d < R;
        const double fd = 1.0 - (inrange ? d : 0.0) / R;
        const double term_d = a.w_d * (fd * fd);
        const double closing = 0.0 - (mx * rvx + my * rvy);
        const bool has_c = (vv > 0.0) && (closing > 0.0);
        const double tcpa = (has_c ? closing : 0.0) / (has_c ? vv : 1.0);
        const double qx = mx + rvx * tcpa;
        const double qy = my + rvy * tcpa;
        const double dcpa = __builtin_sqrt(qx * qx + qy * qy);
        const double fc = 1.0 - (dcpa < R ? dcpa : R) / R;
        const double term_c = (a.w_c * (fc * fc)) * (a.t_c / (a.t_c + tcpa));
        const double rr = term_d + (has_c ? term_c : 0.0);
        a.risk[eo] = a.risk_model == 1 ? risk_reference(mx, my, vx, vy, vox, voy, dref, sc.c, sc.s, psi, dref < __builtin_inf()) : (inrange ? rr : 0.0);
    }
    __syncthreads();
    for (int el = threadIdx.x; el < ne; el += blockDim.x) {
        const int e = e0 + el;
        double best = __builtin_inf();
        int bi = -1;
        for (int o = 0; o < O; ++o) {
            const double d = lds_dist[el * O + o];
            if (d < best) {
                best = d;
                bi = o;
            }
        }
        const uint8_t coll = best < a.contact_dist ? 1 : 0;
        a.min_dist[e] = best;
        a.min_idx[e] = bi;
        a.collision[e] = coll;
        const double px = a.state[e];
        const double py = a.state[(size_t)V + e];
        const double psi = a.state[2 * (size_t)V + e];
        const double u = a.state[3 * (size_t)V + e];
        const double v = a.state[4 * (size_t)V + e];
        const double r = a.state[5 * (size_t)V + e];
        const SinCos sc = sincos_det(psi);
        const double vox = sc.c * u - sc.s * v;
        const double voy = sc.s * u + sc.c * v;
        a.own[2 * (size_t)e] = __builtin_sqrt(vox * vox + voy * voy);
        a.own[2 * (size_t)e + 1] = psi;
        int seg = a.path_seg[e];
        const double* tab = a.path + (size_t)e * a.KS * kPathRec;
        const int nseg = a.path_nseg[e];
        const PathNav pn = path_navigation(tab, nseg, seg, load_path_record(tab, seg), load_path_record(tab, nseg - 1), px,
                                           py, psi, a.look_ahead);
        a.path_seg[e] = seg;
        const double al = pn.along, cte = pn.cte, la = pn.la, he = pn.he, len = pn.total;
        float* nav = a.navigation + (size_t)e * 6;
        nav[0] = (float)u;
        nav[1] = (float)v;
        nav[2] = (float)r;
        nav[3] = (float)la;
        nav[4] = (float)he;
        nav[5] = (float)cte;
        const double al_prev = a.along[e];
        a.along[e] = al;
        const uint8_t goal = ((len - al) <= a.goal_radius) ? 1 : 0;
        const uint8_t trunc = (a.t_step[e] >= a.max_steps) ? 1 : 0;
        double rew = 0.0;
        if (a.is_step) {
            rew = a.rw_progress * (al - al_prev);
            rew = rew - a.rw_cte * __builtin_fabs(cte);
            rew = rew - (coll ? a.rw_collision : 0.0);
            rew = rew + (goal ? a.rw_goal : 0.0);
        }
        a.reward[e] = rew;
        a.reached_goal[e] = goal;
        a.truncated[e] = trunc;
        a.done[e] = coll | goal | trunc;
    }
}

// ---------------------------------------------------------------------------
// LiDAR
// ---------------------------------------------------------------------------
// Squared distance of a polygon's farthest vertex from its pose point, or -1
// unless every vertex lies strictly to the left of every edge it does not
// belong to (strictly convex, counter-clockwise).  v: body-frame vertices,
// nv of them in use.  A convex polygon seen from farther out than that shows
// the sensor only its facing edges: an edge with a x e > 0 (outward side away
// from the vessel) is then no LiDAR surface.  Seen from outside, every ray
// meets a facing edge first, so only rays through a silhouette vertex could
// tell the difference.
__host__ __device__ inline double polygon_reach2(const double* v, int nv) {
    if (nv < 3) return -1.0;
    double reach2 = 0.0;
    for (int k = 0; k < nv; ++k) {
        const int k1 = (k + 1 < nv) ? k + 1 : 0;
        const double ex = v[2 * k1] - v[2 * k], ey = v[2 * k1 + 1] - v[2 * k + 1];
        const double rr = v[2 * k] * v[2 * k] + v[2 * k + 1] * v[2 * k + 1];
        reach2 = rr > reach2 ? rr : reach2;
        for (int j = 0; j < nv; ++j) {
            if (j == k || j == k1) continue;
            const double wx = v[2 * j] - v[2 * k], wy = v[2 * j + 1] - v[2 * k + 1];
            if (!(ex * wy - ey * wx > 0.0)) return -1.0;
        }
    }
    return reach2;
}

struct LidarArgs {
    Dims d;
    int TE;  // envs per workgroup
    const double* state;
    const double* circles;
    const double* poly_pose;
    const double* poly_verts;
    const int32_t* poly_nv;
    const double* poly_reach2;  // [V,P] polygon_reach2
    const double* ray_dirs;  // [R,2]
    double* ranges;          // [V,R]
    int32_t* hit;            // [V,R]
    float* perception;       // [V,1,R]
    double sensor_range;
};

// LDS image per tile: pose[TE][4] = px,py,cos psi,sin psi | circ[TE][C][3] =
// mx,my,cc | seg[TE][S][5] = ax,ay,ex,ey,nt  (S = P*NV), all vessel-relative.
__host__ __device__ inline size_t lidar_lds_doubles(int TE, int C, int S) {
    return (size_t)TE * (4 + 3 * (size_t)C + 5 * (size_t)S);
}

__global__ __launch_bounds__(256) void k_lidar(LidarArgs a) {
    extern __shared__ double lds[];
    const int V = a.d.V, R = a.d.R, C = a.d.C, P = a.d.P, NV = a.d.NV;
    const int S = P * NV;
    const int TE = a.TE;
    double* pose = lds;
    double* circ = pose + (size_t)TE * 4;
    double* seg = circ + (size_t)TE * C * 3;
    const int e0 = blockIdx.x * TE;
    const int ne = min(TE, V - e0);
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;

    for (int el = tid; el < ne; el += nthr) {
        const int e = e0 + el;
        const SinCos sc = sincos_det(a.state[2 * (size_t)V + e]);
        pose[el * 4 + 0] = a.state[e];
        pose[el * 4 + 1] = a.state[(size_t)V + e];
        pose[el * 4 + 2] = sc.c;
        pose[el * 4 + 3] = sc.s;
    }
    for (int i = tid; i < ne * C; i += nthr) {
        const int el = i / C;
        const int e = e0 + el;
        const double* c = a.circles + ((size_t)e0 * C + i) * 5;
        const double rad = c[2];
        const double mx = c[0] - a.state[e];
        const double my = c[1] - a.state[(size_t)V + e];
        const double cc = (mx * mx + my * my) - rad * rad;
        const bool on = rad > 0.0;
        // an absent circle gets cc = +inf: its discriminant is -inf, a miss
        circ[i * 3 + 0] = on ? mx : 0.0;
        circ[i * 3 + 1] = on ? my : 0.0;
        circ[i * 3 + 2] = on ? cc : __builtin_inf();
    }
    for (int i = tid; i < ne * S; i += nthr) {
        const int el = i / S;
        const int s = i - el * S;
        const int q = s / NV;
        const int k = s - q * NV;
        const int e = e0 + el;
        const size_t ep = (size_t)e * P + q;
        const int nv = a.poly_nv[ep];
        double ax = 0.0, ay = 0.0, ex = 0.0, ey = 0.0, nt = 0.0;
        if (k < nv) {
            // an unused edge slot stays all-zero: its denominator is 0, a miss
            const double* p = a.poly_pose + ep * 6;
            const double* vb = a.poly_verts + ep * NV * 2;
            const int k1 = (k + 1 < nv) ? k + 1 : 0;
            const double X = p[0], Y = p[1], ch = p[2], sh = p[3];
            const double Ax = X + (ch * vb[2 * k] - sh * vb[2 * k + 1]);
            const double Ay = Y + (sh * vb[2 * k] + ch * vb[2 * k + 1]);
            const double Bx = X + (ch * vb[2 * k1] - sh * vb[2 * k1 + 1]);
            const double By = Y + (sh * vb[2 * k1] + ch * vb[2 * k1 + 1]);
            ax = Ax - a.state[e];
            ay = Ay - a.state[(size_t)V + e];
            ex = Bx - Ax;
            ey = By - Ay;
            nt = ax * ey - ay * ex;
            // an edge of a convex polygon that faces away from a vessel outside it is no LiDAR surface:
            // its record becomes the all-zero one of an unused slot
            const double reach2 = a.poly_reach2[ep];
            const double mx = X - a.state[e], my = Y - a.state[(size_t)V + e];
            if (reach2 >= 0.0 && (mx * mx + my * my) > reach2 && nt > 0.0) {
                ax = 0.0;
                ay = 0.0;
                ex = 0.0;
                ey = 0.0;
                nt = 0.0;
            }
        }
        double* sg = seg + (size_t)i * 5;
        sg[0] = ax;
        sg[1] = ay;
        sg[2] = ex;
        sg[3] = ey;
        sg[4] = nt;
    }
    __syncthreads();

    const double range = a.sensor_range;
    for (int f = tid; f < ne * R; f += nthr) {
        const int el = f / R;
        const int r = f - el * R;
        const double cpsi = pose[el * 4 + 2];
        const double spsi = pose[el * 4 + 3];
        const double cb = a.ray_dirs[2 * r];
        const double sb = a.ray_dirs[2 * r + 1];
        const double dx = cpsi * cb - spsi * sb;
        const double dy = spsi * cb + cpsi * sb;
        double best = range;
        int hit = -1;
        const double* cp = circ + (size_t)el * C * 3;
        for (int j = 0; j < C; ++j) {
            const double mx = cp[3 * j], my = cp[3 * j + 1], cc = cp[3 * j + 2];
            const double b = mx * dx + my * dy;
            const double disc = b * b - cc;
            if (disc >= 0.0) {
                const double sq = __builtin_sqrt(disc);
                double t = b - sq;
                t = t < 0.0 ? b + sq : t;
                if (t >= 0.0 && t < best) {
                    best = t;
                    hit = j;
                }
            }
        }
        const double* sp = seg + (size_t)el * S * 5;
        int s = 0;
        for (int q = 0; q < P; ++q) {
            for (int k = 0; k < NV; ++k, ++s) {
                const double ax = sp[5 * s], ay = sp[5 * s + 1], ex = sp[5 * s + 2], ey = sp[5 * s + 3];
                const double den = dx * ey - dy * ex;
                const double ns = ax * dy - ay * dx;
                const bool inseg = den > 0.0 ? (ns >= 0.0 && ns <= den) : (ns <= 0.0 && ns >= den);
                if (den != 0.0 && inseg) {
                    const double t = sp[5 * s + 4] / den;
                    if (t >= 0.0 && t < best) {
                        best = t;
                        hit = C + q;
                    }
                }
            }
        }
        const size_t out = (size_t)(e0 + el) * R + r;
        a.ranges[out] = best;
        a.hit[out] = hit;
        a.perception[out] = (float)(1.0 - best / range);
    }
}

}  // namespace auv
