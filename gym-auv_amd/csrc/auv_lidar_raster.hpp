// Work-efficient LiDAR: instead of testing every ray against every primitive
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
// margin (0.02 ray spacings, >= 1e-4 rad for any scan this library accepts)
// that is orders of magnitude above both the fp32 error (~1e-6 rad) and the
// deviation between the fp64 predicates and exact geometry (<= ~1e-8 rad), so
// no pair that the exact test would accept is ever skipped.
#pragma once
#include "auv_kernels.hpp"

namespace auv {

struct RasterArgs {
    LidarArgs l;
    int cap;  // pair-list slots per round (multiple of 256)
};

struct RasterLayout {
    size_t pose, circ, seg, best, meta, offs, ids, pairs, wsum, total;  // byte offsets
};

// LDS image per tile.  Primitive p of the tile: circles of all its envs first
// (p = el*C + j), then edges (p = TE*C + el*S + s).
//   pose[TE][4] f64 | circ[TE*C][3] f64 | seg[TE*S][5] f64 | best[TE*R] u64
//   | meta[NP] int4 {first ray, count, env in tile, obstacle id} | offs[NP+1] i32
//   | ids[TE*R] u32 | pairs[cap] u32 {p << 16 | k} | wsum[8] i32
__host__ __device__ inline RasterLayout raster_layout(int TE, int R, int C, int S, int cap) {
    RasterLayout L;
    const size_t NP = (size_t)TE * ((size_t)C + S);
    size_t o = 0;
    L.pose = o; o += (size_t)TE * 4 * 8;
    L.circ = o; o += (size_t)TE * C * 3 * 8;
    L.seg = o; o += (size_t)TE * S * 5 * 8;
    L.best = o; o += (size_t)TE * R * 8;
    o = (o + 15) & ~(size_t)15;
    L.meta = o; o += NP * 16;
    L.offs = o; o += (NP + 1) * 4;
    L.ids = o; o += (size_t)TE * R * 4;
    L.pairs = o; o += (size_t)cap * 4;
    L.wsum = o; o += 8 * 4;
    L.total = (o + 15) & ~(size_t)15;
    return L;
}

constexpr float kIvMargin = 0.02f;      // ray spacings added on each side
constexpr float kRangeSlack = 1.0001f;  // relative slack of the range cull

// fractional scan index of a world bearing: ray i sits at x = i, x in [-1, R-1)
__device__ __forceinline__ float scan_index(float bearing, float psi, float inv_delta) {
    float w = bearing - psi;
    const float two_pi = 6.2831853f, pi = 3.14159265f;
    w = w - two_pi * floorf((w + pi) * (1.0f / two_pi));
    return (w + pi) * inv_delta - 1.0f;
}

__device__ __forceinline__ int2 make_interval(float lo, float hi, int R) {
    const int ilo = (int)floorf(lo - kIvMargin);
    const int ihi = (int)ceilf(hi + kIvMargin);
    int cnt = ihi - ilo + 1;
    int first = ilo % R;
    if (first < 0) first += R;
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

// one sweep over the dense pair list of the current round
template <int PASS>
__device__ __forceinline__ void raster_sweep(const RasterArgs& a, int n_pairs, int n_circ, const double* pose,
                                             const double* circ, const double* seg, unsigned long long* best,
                                             const int4* meta, const unsigned int* pairs, unsigned int* ids) {
    const int R = a.l.d.R;
    const double range = a.l.sensor_range;
    for (int w = threadIdx.x; w < n_pairs; w += blockDim.x) {
        const unsigned int pr = pairs[w];
        const int p = (int)(pr >> 16);
        const int4 m = meta[p];
        int r = m.x + (int)(pr & 0xffffu);
        r = r >= R ? r - R : r;
        const double cpsi = pose[m.z * 4 + 2], spsi = pose[m.z * 4 + 3];
        const double cb = a.l.ray_dirs[2 * r], sb = a.l.ray_dirs[2 * r + 1];
        const double dx = cpsi * cb - spsi * sb;
        const double dy = spsi * cb + cpsi * sb;
        double t;
        bool ok;
        if (p < n_circ) {
            ok = circle_hit(circ[3 * p], circ[3 * p + 1], circ[3 * p + 2], dx, dy, t);
        } else {
            const double* sg = seg + (size_t)(p - n_circ) * 5;
            ok = edge_hit(sg[0], sg[1], sg[2], sg[3], sg[4], dx, dy, t);
        }
        if (ok && t < range) {
            const unsigned long long tb = t_key(t);
            const int slot = m.z * R + r;
            if (PASS == 0) atomicMin(&best[slot], tb);
            else if (best[slot] == tb) atomicMin(&ids[slot], (unsigned int)m.w);
        }
    }
}

__global__ __launch_bounds__(256) void k_lidar_raster(RasterArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int V = a.l.d.V, R = a.l.d.R, C = a.l.d.C, P = a.l.d.P, NV = a.l.d.NV;
    const int S = P * NV;
    const int TE = a.l.TE;
    const RasterLayout L = raster_layout(TE, R, C, S, a.cap);
    double* pose = reinterpret_cast<double*>(lds_raw + L.pose);
    double* circ = reinterpret_cast<double*>(lds_raw + L.circ);
    double* seg = reinterpret_cast<double*>(lds_raw + L.seg);
    unsigned long long* best = reinterpret_cast<unsigned long long*>(lds_raw + L.best);
    int4* meta = reinterpret_cast<int4*>(lds_raw + L.meta);
    int* offs = reinterpret_cast<int*>(lds_raw + L.offs);
    unsigned int* ids = reinterpret_cast<unsigned int*>(lds_raw + L.ids);
    unsigned int* pairs = reinterpret_cast<unsigned int*>(lds_raw + L.pairs);
    int* wsum = reinterpret_cast<int*>(lds_raw + L.wsum);
    const int e0 = blockIdx.x * TE;
    const int ne = min(TE, V - e0);
    const int n_circ = ne * C;
    const int NP = n_circ + ne * S;
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const double range = a.l.sensor_range;
    const float frange = (float)range * kRangeSlack;
    const float inv_delta = (float)R * (1.0f / 6.2831853f);
    const float half_scan = 0.5f * (float)R;

    // ---- stage: pose, per-ray accumulators, vessel-relative primitives + intervals
    for (int el = tid; el < ne; el += nthr) {
        const int e = e0 + el;
        const SinCos sc = sincos_det(a.l.state[2 * (size_t)V + e]);
        pose[el * 4 + 0] = a.l.state[e];
        pose[el * 4 + 1] = a.l.state[(size_t)V + e];
        pose[el * 4 + 2] = sc.c;
        pose[el * 4 + 3] = sc.s;
    }
    const unsigned long long range_bits = (unsigned long long)__double_as_longlong(range);
    for (int f = tid; f < ne * R; f += nthr) {
        best[f] = range_bits;
        ids[f] = 0xffffffffu;
    }
    for (int i = tid; i < n_circ; i += nthr) {
        const int el = i / C;
        const int j = i - el * C;
        const int e = e0 + el;
        const double* c = a.l.circles + ((size_t)e0 * C + i) * 5;
        const double rad = c[2];
        const double mx = c[0] - a.l.state[e];
        const double my = c[1] - a.l.state[(size_t)V + e];
        const double cc = (mx * mx + my * my) - rad * rad;
        const bool on = rad > 0.0;
        circ[i * 3 + 0] = on ? mx : 0.0;
        circ[i * 3 + 1] = on ? my : 0.0;
        circ[i * 3 + 2] = on ? cc : __builtin_inf();
        const float fr = (float)rad, fmx = (float)mx, fmy = (float)my;
        const float mm = fmx * fmx + fmy * fmy;
        int2 v = make_int2(0, 0);
        if (!on) {
            // absent slot
        } else if (!(mm > fr * fr * 1.001f)) {
            v.y = R;  // vessel inside or touching: every ray can hit
        } else if (sqrtf(mm) - fr > frange) {
            // wholly beyond sensor range
        } else {
            const float psi = (float)a.l.state[2 * (size_t)V + e];
            const float xc = scan_index(atan2f(fmy, fmx), psi, inv_delta);
            const float hw = atan2f(fr, sqrtf(mm - fr * fr)) * inv_delta;
            v = make_interval(xc - hw, xc + hw, R);
        }
        meta[i] = make_int4(v.x, v.y, el, j);
    }
    for (int i = tid; i < ne * S; i += nthr) {
        const int el = i / S;
        const int s = i - el * S;
        const int q = s / NV;
        const int k = s - q * NV;
        const int e = e0 + el;
        const size_t ep = (size_t)e * P + q;
        const int nv = a.l.poly_nv[ep];
        double ax = 0.0, ay = 0.0, ex = 0.0, ey = 0.0, nt = 0.0;
        int2 v = make_int2(0, 0);
        if (k < nv) {
            const double* p = a.l.poly_pose + ep * 6;
            const double* vb = a.l.poly_verts + ep * NV * 2;
            const int k1 = (k + 1 < nv) ? k + 1 : 0;
            const double X = p[0], Y = p[1], ch = p[2], sh = p[3];
            const double Ax = X + (ch * vb[2 * k] - sh * vb[2 * k + 1]);
            const double Ay = Y + (sh * vb[2 * k] + ch * vb[2 * k + 1]);
            const double Bx = X + (ch * vb[2 * k1] - sh * vb[2 * k1 + 1]);
            const double By = Y + (sh * vb[2 * k1] + ch * vb[2 * k1 + 1]);
            ax = Ax - a.l.state[e];
            ay = Ay - a.l.state[(size_t)V + e];
            ex = Bx - Ax;
            ey = By - Ay;
            nt = ax * ey - ay * ex;
            // candidate rays: the arc between the bearings of the two endpoints
            const float fax = (float)ax, fay = (float)ay, fex = (float)ex, fey = (float)ey;
            const float fbx = fax + fex, fby = fay + fey;
            const float ee = fex * fex + fey * fey;
            float tp = ee > 0.0f ? -(fax * fex + fay * fey) / ee : 0.0f;
            tp = fminf(fmaxf(tp, 0.0f), 1.0f);
            const float qx = fax + tp * fex, qy = fay + tp * fey;
            const float dmin = sqrtf(qx * qx + qy * qy);
            const float aa = fax * fax + fay * fay, bb = fbx * fbx + fby * fby;
            if (dmin > frange) {
                // wholly beyond sensor range: stays empty
            } else if (aa < 1e-6f || bb < 1e-6f || dmin < 1e-3f) {
                v.y = R;  // an endpoint (or the edge) is on top of the vessel: bearings are meaningless
            } else {
                const float psi = (float)a.l.state[2 * (size_t)V + e];
                const float xa = scan_index(atan2f(fay, fax), psi, inv_delta);
                const float xb = scan_index(atan2f(fby, fbx), psi, inv_delta);
                float d = xb - xa;
                d = d > half_scan ? d - (float)R : d;
                d = d < -half_scan ? d + (float)R : d;
                if (fabsf(d) > half_scan - 1.5f) {
                    v.y = R;  // subtends ~180 deg: the short way round is ambiguous
                } else {
                    const float lo = d >= 0.0f ? xa : xa + d;
                    v = make_interval(lo, lo + fabsf(d), R);
                }
            }
        }
        double* sg = seg + (size_t)i * 5;
        sg[0] = ax;
        sg[1] = ay;
        sg[2] = ex;
        sg[3] = ey;
        sg[4] = nt;
        meta[n_circ + i] = make_int4(v.x, v.y, el, C + q);
    }
    __syncthreads();

    // ---- exclusive prefix sum of the interval lengths: offs[p], offs[NP] = total
    const int chunk = (NP + nthr - 1) / nthr;
    const int p_lo = min(tid * chunk, NP), p_hi = min(p_lo + chunk, NP);
    int mine = 0;
    for (int p = p_lo; p < p_hi; ++p) mine += meta[p].y;
    int incl = mine;
    const int lane = tid & 63;
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int wv = 0; wv < (tid >> 6); ++wv) base += wsum[wv];
    int run = base + incl - mine;
    for (int p = p_lo; p < p_hi; ++p) {
        offs[p] = run;
        run += meta[p].y;
    }
    int total = 0;
    for (int wv = 0; wv < (nthr >> 6); ++wv) total += wsum[wv];
    __syncthreads();

    // ---- sweeps: rounds of at most `cap` pairs (one round in the common case)
    const int n_rounds = (total + a.cap - 1) / a.cap;
    for (int pass = 0; pass < 2; ++pass) {
        for (int rd = 0; rd < n_rounds; ++rd) {
            const int w0 = rd * a.cap;
            const int n_pairs = min(a.cap, total - w0);
            if (pass == 0 || n_rounds > 1) {
                for (int p = tid; p < NP; p += nthr) {
                    const int o = offs[p];
                    const int lo = max(o, w0), hi = min(o + meta[p].y, w0 + a.cap);
                    for (int w = lo; w < hi; ++w) pairs[w - w0] = ((unsigned int)p << 16) | (unsigned int)(w - o);
                }
                __syncthreads();
            }
            if (pass == 0) raster_sweep<0>(a, n_pairs, n_circ, pose, circ, seg, best, meta, pairs, ids);
            else raster_sweep<1>(a, n_pairs, n_circ, pose, circ, seg, best, meta, pairs, ids);
            __syncthreads();
        }
    }

    // ---- output: 16 B per ray, coalesced
    for (int f = tid; f < ne * R; f += nthr) {
        const double t = __longlong_as_double((long long)best[f]);
        const size_t out = (size_t)e0 * R + f;
        a.l.ranges[out] = t;
        a.l.hit[out] = (int)ids[f];
        a.l.perception[out] = (float)(1.0 - t / range);
    }
}

}  // namespace auv
