// Scenario generation on the device: static circles and constant-velocity
// ship polygons scattered round a vessel at the origin, one straight path per
// env.  Every number is a function of (seed, env, episode, purpose, index)
// through Philox4x32-10 (Salmon et al., SC'11), so one env's scene can be
// regenerated for its next episode without touching any other env, and a numpy
// restatement reproduces the bits.  Doubles carry 53 random bits exactly;
// arithmetic is unfused fp64 in one fixed order; sin/cos are sincos_det.
#pragma once
#include "auv_kernels.hpp"

namespace auv {

struct Philox4 {
    uint32_t x, y, z, w;
};

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

enum ScenePurpose { SG_VESSEL = 0, SG_CIRCLE_POS, SG_CIRCLE_RAD, SG_POLY_POS, SG_POLY_MOTION, SG_POLY_WIDTH, SG_POLY_SPEED };

struct Uniform2 {
    double a, b;
};

// two doubles in [0,1), 53 bits each, from one block
__device__ __forceinline__ Uniform2 uniform2(uint64_t seed, int env, int episode, int purpose, int index) {
    const Philox4 r = philox4x32_10((uint32_t)env, (uint32_t)episode, (uint32_t)purpose, (uint32_t)index,
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t m0 = ((uint64_t)r.x << 21) | (r.y >> 11);
    const uint64_t m1 = ((uint64_t)r.z << 21) | (r.w >> 11);
    const double s = 1.0 / 9007199254740992.0;
    return Uniform2{(double)m0 * s, (double)m1 * s};
}

struct SceneGenArgs {
    Dims d;
    uint64_t seed;
    // null: every env, for the episode number it has; else only envs whose flag
    // is set, for their NEXT episode (the counter itself is advanced by k_vessel
    // when it restarts the env, later in the same step)
    const uint8_t* only_done;
    const int32_t* episode;  // [V]
    double* circles0;           // [V,C,5]
    double* poly_pose0;         // [V,P,6]
    double* poly_verts;         // [V,P,NV,2]
    int32_t* poly_nv;
    double* poly_heading;
    double* poly_beam;
    double* poly_reach2;
    double* poly_rot;           // [V,P,NV,2]
    double* path;  // [V,KS,kPathRec]: the generator lays one straight segment
    int32_t* path_nseg;
    int KS;
    double* init;  // [6,V]
    double extent, r_lo, r_hi, w_lo, w_hi, s_lo, s_hi, clearance, path_length;
    int size_law;  // 0: uniform; 1: whole-number sizes and speeds with a triangular law
};

// a size in [lo, hi]: uniform from u, or (bell) a whole number lo..hi from the mean of two uniforms
__device__ __forceinline__ double draw_size(double lo, double hi, Uniform2 u, int bell) {
    if (bell) return __builtin_floor(lo + ((hi + 1.0) - lo) * ((u.a + u.b) * 0.5));
    return lo + (hi - lo) * u.a;
}

__device__ __forceinline__ void push_out(double& x, double& y, double clear) {
    double d = __builtin_sqrt(x * x + y * y);
    d = d == 0.0 ? 1.0 : d;
    const double f = d < clear ? clear / d : 1.0;
    x = x * f;
    y = y * f;
}

// one thread per (env, slot): slot 0 = vessel + path, then the circles, then the polygons
__global__ __launch_bounds__(256) void k_scenegen(SceneGenArgs a) {
    const int V = a.d.V, C = a.d.C, P = a.d.P, NV = a.d.NV;
    const int slots = 1 + C + P;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)V * slots) return;
    const int e = (int)(gid / slots);
    const int slot = (int)(gid - (long long)e * slots);
    int ep = a.episode[e];
    if (a.only_done != nullptr) {
        if (a.only_done[e] == 0) return;
        ep = ep + 1;
    }
    const double E = a.extent;
    if (slot == 0) {
        const Uniform2 u = uniform2(a.seed, e, ep, SG_VESSEL, 0);
        const double psi = u.a * kTwoPi - kPi;
        const SinCos sc = sincos_det(psi);
        for (int f = 0; f < 6; ++f) a.init[(size_t)f * V + e] = f == 2 ? psi : 0.0;
        double* pt = a.path + (size_t)e * a.KS * kPathRec;
        pt[0] = 0.0;
        pt[1] = 0.0;
        pt[2] = psi;
        pt[3] = sc.c;
        pt[4] = sc.s;
        pt[5] = a.path_length;
        pt[6] = 0.0;
        pt[7] = 0.0;
        a.path_nseg[e] = 1;
    } else if (slot <= C) {
        const int j = slot - 1;
        const Uniform2 up = uniform2(a.seed, e, ep, SG_CIRCLE_POS, j);
        const Uniform2 ur = uniform2(a.seed, e, ep, SG_CIRCLE_RAD, j);
        const double rad = draw_size(a.r_lo, a.r_hi, ur, a.size_law);
        double cx = up.a * (2.0 * E) - E;
        double cy = up.b * (2.0 * E) - E;
        push_out(cx, cy, rad + a.clearance);
        double* c = a.circles0 + ((size_t)e * C + j) * 5;
        c[0] = cx;
        c[1] = cy;
        c[2] = rad;
        c[3] = 0.0;
        c[4] = 0.0;
    } else {
        const int q = slot - 1 - C;
        const Uniform2 up = uniform2(a.seed, e, ep, SG_POLY_POS, q);
        const Uniform2 um = uniform2(a.seed, e, ep, SG_POLY_MOTION, q);
        const Uniform2 uw = uniform2(a.seed, e, ep, SG_POLY_WIDTH, q);
        const double w = draw_size(a.w_lo, a.w_hi, uw, a.size_law);
        double spd = a.s_lo + (a.s_hi - a.s_lo) * um.b;
        if (a.size_law) {
            const Uniform2 us = uniform2(a.seed, e, ep, SG_POLY_SPEED, q);
            spd = a.s_lo + (a.s_hi - a.s_lo) * ((us.a + us.b) * 0.5);
        }
        const double h = um.a * kTwoPi - kPi;
        double X = up.a * (2.0 * E) - E;
        double Y = up.b * (2.0 * E) - E;
        push_out(X, Y, 2.0 * w + a.clearance);
        const SinCos sc = sincos_det(h);
        const size_t ep_i = (size_t)e * P + q;
        double* p = a.poly_pose0 + ep_i * 6;
        p[0] = X;
        p[1] = Y;
        p[2] = sc.c;
        p[3] = sc.s;
        p[4] = spd * sc.c;
        p[5] = spd * sc.s;
        const double hw = w * 0.5;
        double* v = a.poly_verts + ep_i * NV * 2;
        // ship outline: square stern, pointed bow, counter-clockwise (NV >= 5 is checked on the host)
        v[0] = -w;  v[1] = -hw;
        v[2] = 0.0; v[3] = -hw;
        v[4] = w;   v[5] = 0.0;
        v[6] = 0.0; v[7] = hw;
        v[8] = -w;  v[9] = hw;
        for (int k = 10; k < 2 * NV; ++k) v[k] = 0.0;
        a.poly_nv[ep_i] = 5;
        a.poly_heading[ep_i] = h;
        a.poly_beam[ep_i] = hw - (-hw);
        a.poly_reach2[ep_i] = polygon_reach2(v, 5);
        double* rt = a.poly_rot + ep_i * NV * 2;
        for (int k = 0; k < NV; ++k) {
            rt[2 * k] = sc.c * v[2 * k] - sc.s * v[2 * k + 1];
            rt[2 * k + 1] = sc.s * v[2 * k] + sc.c * v[2 * k + 1];
        }
    }
}

}  // namespace auv
