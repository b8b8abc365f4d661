// The step as two launches: a vessel-centric kernel and a scene-centric kernel.
//
// k_vessel   one thread per vessel: RKF45 step (skipped after a reset), then
//            everything that depends on the vessel and its path only: heading
//            sin/cos, own velocity, navigation features, along-track progress,
//            goal / truncation flags, the scene-independent part of the reward.
// k_sense    one workgroup per tile of envs, reading their scene ONCE from HBM:
//   1. one thread per primitive (circle / polygon edge): (step only) advances
//      the constant-velocity obstacle and writes it back; makes the primitive
//      vessel-relative in LDS; per edge also the squared distance to the vessel
//      and the crossing-number bit.  Edges that a ray can reach (in range,
//      facing the vessel) are listed per block of 64 edges, each block with a
//      count of its own; there is no pair list and no shared counter;
//   2. per obstacle, as two tasks on two waves: signed hull distance with the
//      per-env summary (nearest obstacle, collision, reward, done); bearing,
//      course, relative speed and risk;
//   3. LiDAR, in the same barrier interval as 2, on the other waves: the circles
//      and the listed edges are swept in units of 64 primitives; the unit's wave
//      derives each primitive's candidate ray interval (auv_lidar_raster.hpp),
//      deals the unit's (primitive, ray) pairs evenly over its lanes and runs
//      the exact fp64 test of each; LDS atomic min per ray, then id resolution;
//   4. 16 B per ray written out; with a network attached, its forward pass on
//      the closeness values still in LDS.
// Results are bit-identical to k_dynamics(advance) + k_encounter + k_lidar
// (+ k_cnn).
#pragma once
#include "auv_cnn.hpp"
#include "auv_lidar_raster.hpp"
#include "auv_risk.hpp"

namespace auv {

typedef unsigned long long auv_u64x2 __attribute__((ext_vector_type(2)));
typedef unsigned int auv_u32x4 __attribute__((ext_vector_type(4)));
typedef double auv_f64x2 __attribute__((ext_vector_type(2)));
typedef float auv_f32x4 __attribute__((ext_vector_type(4)));

// per-vessel record handed from k_vessel to k_sense, f64 [V,8]
enum EnvRec { ER_COS = 0, ER_SIN, ER_VOX, ER_VOY, ER_REWARD, ER_RESET, ER_COUNT = 8 };

struct VesselArgs {
    int V;
    int do_step;  // 1: integrate and count the step; 0: state was just (re)set
    int auto_reset;  // 1: an env whose done flag is set restarts from `init` instead of integrating
    int max_steps;
    const uint8_t* done;  // flags of the previous sensing pass
    const double* init;   // [6,V] initial states of the last reset call
    // the obstacles of a restarting env go back to their initial records
    // (uploaded, or just regenerated for the new episode); episode[] counts restarts
    int32_t* episode;
    int C, P;
    double* circles;           // [V,C,5]
    const double* circles0;
    double* poly_pose;         // [V,P,6]
    const double* poly_pose0;
    double* state;         // [6,V]
    const double* action;  // [2,V]
    int32_t* t_step;
    const double* path;  // [V,KS,kPathRec]
    const int32_t* path_nseg;  // [V] segments in use
    int32_t* path_seg;         // [V] the segment each vessel is on
    int KS;
    double* envrec;      // [V,8]
    double* own;         // [V,2]
    double* along;       // [V]
    float* navigation;   // [V,1,6]
    uint8_t* reached_goal;
    uint8_t* truncated;
    double dt, thrust_max, moment_max, look_ahead, goal_radius, rw_progress, rw_cte;
    double mp[MP_COUNT];
};

__global__ __launch_bounds__(256) void k_vessel(VesselArgs a) {
    const int V = a.V;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < V;  // no early exit: the whole workgroup takes part in the restore below
    // Envs that restart get their obstacle records back from the initial ones.
    // The workgroup shares that copy, and its loads are issued here, before the
    // integration, whose ~8 us hide their latency; the stores follow at the end.
    __shared__ int rs_env[256];
    __shared__ int rs_n;
    if (threadIdx.x == 0) rs_n = 0;
    __syncthreads();
    bool restart = live && a.do_step && a.auto_reset && a.done[i] != 0;
    if (restart) rs_env[atomicAdd(&rs_n, 1)] = i;  // at most one entry per thread of the 256
    __syncthreads();
    const int n_el = a.C * 5, n_rec = a.C * 5 + a.P * 6;  // doubles per env: circles, then polygon poses
    const int n_items = rs_n * n_rec;
    auto item_src = [&](int item) -> const double* {
        const int r = item / n_rec, k = item - r * n_rec;
        const size_t e = (size_t)rs_env[r];
        return k < n_el ? a.circles0 + e * n_el + k : a.poly_pose0 + e * (n_rec - n_el) + (k - n_el);
    };
    auto item_dst = [&](int item) -> double* {
        const int r = item / n_rec, k = item - r * n_rec;
        const size_t e = (size_t)rs_env[r];
        return k < n_el ? a.circles + e * n_el + k : a.poly_pose + e * (n_rec - n_el) + (k - n_el);
    };
    const int nthr = blockDim.x;
    double held[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((int)threadIdx.x + j * nthr < n_items) held[j] = *item_src(threadIdx.x + j * nthr);
    if (live) {
    // the path records this vessel will need, asked for ahead of the integration
    const double* tab = a.path + (size_t)i * a.KS * kPathRec;
    const int nseg = a.path_nseg[i];
    int seg = restart ? 0 : a.path_seg[i];
    const PathRecord cur = load_path_record(tab, seg), last = load_path_record(tab, nseg - 1);
    Vec6 s = load_state(a.state, V, i);
    int t = a.t_step[i];
    if (a.do_step) {
        if (restart) {
            s = load_state(a.init, V, i);
            t = 0;
            a.episode[i] = a.episode[i] + 1;
        } else {
            s = rkf45_step(s, a.action[i], a.action[(size_t)V + i], a.dt, a.thrust_max, a.moment_max, a.mp);
            t = t + 1;
        }
        store_state(a.state, V, i, s);
        a.t_step[i] = t;
    }
    const SinCos sc = sincos_det(s.psi);
    const double vox = sc.c * s.u - sc.s * s.v;
    const double voy = sc.s * s.u + sc.c * s.v;
    a.own[2 * (size_t)i] = __builtin_sqrt(vox * vox + voy * voy);
    a.own[2 * (size_t)i + 1] = s.psi;
    const PathNav pn = path_navigation(tab, nseg, seg, cur, last, s.x, s.y, s.psi, a.look_ahead);
    a.path_seg[i] = seg;
    const double al = pn.along, cte = pn.cte, la = pn.la, he = pn.he, len = pn.total;
    float* nav = a.navigation + (size_t)i * 6;
    nav[0] = (float)s.u;
    nav[1] = (float)s.v;
    nav[2] = (float)s.r;
    nav[3] = (float)la;
    nav[4] = (float)he;
    nav[5] = (float)cte;
    const double al_prev = a.along[i];
    a.along[i] = al;
    a.reached_goal[i] = ((len - al) <= a.goal_radius) ? 1 : 0;
    a.truncated[i] = (t >= a.max_steps) ? 1 : 0;
    double rew = 0.0;
    if (a.do_step && !restart) {
        rew = a.rw_progress * (al - al_prev);
        rew = rew - a.rw_cte * __builtin_fabs(cte);
    }
    double* er = a.envrec + (size_t)i * ER_COUNT;
    er[ER_COS] = sc.c;
    er[ER_SIN] = sc.s;
    er[ER_VOX] = vox;
    er[ER_VOY] = voy;
    er[ER_REWARD] = rew;
    er[ER_RESET] = restart ? 1.0 : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if ((int)threadIdx.x + j * nthr < n_items) *item_dst(threadIdx.x + j * nthr) = held[j];
    for (int item = threadIdx.x + 4 * nthr; item < n_items; item += nthr) *item_dst(item) = *item_src(item);
}

struct SenseArgs {
    Dims d;
    int TE;       // envs per workgroup
    int lds[10];  // byte offsets of the LDS arrays (sense_layout, computed once on the host): env, circ, seg,
                  // best, obst, edist, ids, ecross, live, wsum (the queue follows wsum: sense_queue_offset)
    int is_step;  // 1: advance obstacles, finish the reward; 0: after reset / plain sense
    int with_meta;  // 0: bearing, course and risk of the obstacles are not refreshed by this pass
    int refresh_heading;  // 0: the heading_ts column (a constant of the scene) is rewritten only for restarting envs
    const double* state;
    const double* envrec;
    double* circles;    // [V,C,5]  positions updated in place when is_step
    double* poly_pose;  // [V,P,6]
    const double* poly_verts;
    const int32_t* poly_nv;
    const double* poly_heading;
    const double* poly_beam;  // [V,P]
    const double* poly_reach2;  // [V,P] polygon_reach2: which polygons show only their facing edges, and from how far out
    const double* poly_rot;   // [V,P,NV,2] R(heading) * body-frame vertex
    const double* ray_dirs;
    double* ranges;
    int32_t* hit;
    float* perception;
    double* obst_dist;
    double* risk;
    double* meta;
    uint8_t* collision;
    const uint8_t* reached_goal;
    const uint8_t* truncated;
    uint8_t* done;
    double* min_dist;
    int32_t* min_idx;
    double* reward;
    double dt, sensor_range, vessel_radius;
    double contact_dist;  // collision <=> nearest obstacle distance below this
    double inv_range;  // 1 / sensor_range, rounded to nearest
    double w_d, w_c, t_c;
    double rw_collision, rw_goal;
    // LiDAR-CNN evaluated in the epilogue (cnn_w == nullptr: none); needs
    // R == kCnnLen and cnn_fits_sense_kernel(cnn_kind)
    const float* cnn_w;
    float* cnn_out;  // [V, cnn_out_dim]
    int cnn_kind, cnn_head;
};

struct SenseLayout {
    size_t env, circ, seg, best, obst, edist, ids, ecross, live, wsum, queue, total;  // byte offsets
};

// LDS image per tile (f64 blocks first, then words and bytes).
//   env[TE][2] cos psi, sin psi, then [TE] psi      circ[TE*C][4] mx,my,cc,radius      seg[TE*S][5] ax,ay,ex,ey,nt
//   best[TE*R] u64     obst: 5 fields [TE*O] mx,my,rvx,rvy,dist then 3 fields [TE*P] x,y,heading
//                      (field-major: the lanes of a wave walk the banks)
//   edist[TE*S] f64    ids[TE*R]   ecross[TE*S]   live[TE*S] numbers of the edges that have candidate rays,
//   by blocks of 64 edges   wsum[blocks] how many each block lists
//   queue (general shape only): the two counters the waves draw their work items from, then a byte per item
// With a network evaluated in-kernel (cnn_kind >= 0) its input (R floats per env)
// and its weights reuse the bytes of obst[], and its two activation buffers
// per env those of best[], both dead by then; the regions are padded when
// they are smaller than that.
// packed: the narrow image of the compile-time shape (DefaultShape): edges without their stored cross
// product, 8-bit crossing records and edge numbers.
// The sweep parts a unit's rounds are dealt into at most, and the work items of a tile at most: two tasks per
// round of 64 obstacles, and that many parts per unit of 64 circles or 64 edges.
constexpr int kShareMax = 4;
__host__ __device__ inline size_t sense_max_items(int TE, int C, int P, int NV) {
    const size_t rounds = ((size_t)TE * (C + P) + 63) / 64, units = ((size_t)TE * C + 63) / 64 + ((size_t)TE * P * NV + 63) / 64;
    return 2 * (rounds ? rounds : 1) + units * kShareMax;
}

// the queue's place in the LDS image: right behind wsum[] (one count per block of 64 edges, and one to spare)
__host__ __device__ inline size_t sense_queue_offset(size_t wsum, int TE, int S) {
    return wsum + (((size_t)TE * S + 63) / 64 + 1) * 4;
}

__host__ __device__ inline SenseLayout sense_layout(int TE, int R, int C, int P, int NV, int cnn_kind, bool packed = false) {
    SenseLayout L;
    const size_t S = (size_t)P * NV, O = (size_t)C + P;
    size_t o = 0;
    L.env = o; o += (size_t)TE * 3 * 8;
    L.circ = o; o += (size_t)TE * C * 4 * 8;
    L.seg = o; o += (size_t)TE * S * (packed ? 4 : 5) * 8;
    const size_t act = cnn_kind < 0 ? 0 : (size_t)TE * cnn_act_floats(cnn_kind) * 4;
    const size_t xw = cnn_kind < 0 ? 0 : (((size_t)TE * R + cnn_n_weights(cnn_kind)) * 4 + 15) & ~(size_t)15;
    const size_t bb = (size_t)TE * R * 8, ob = ((size_t)TE * O * 5 + (size_t)TE * P * 3) * 8;
    // best[], ids[] and the network's input (in obst[]) are walked in 16-byte pieces: they start at multiples of 16
    o = (o + 15) & ~(size_t)15;
    L.best = o; o += bb > act ? bb : act;
    o = (o + 15) & ~(size_t)15;
    L.obst = o; o += ob > xw ? ob : xw;
    L.edist = o; o += (size_t)TE * S * 8;
    o = (o + 15) & ~(size_t)15;
    L.ids = o; o += (size_t)TE * R * 4;
    L.ecross = o; o += (size_t)TE * S * (packed ? 1 : 4);
    L.live = o; o += (size_t)TE * S * (packed ? 1 : 2);
    o = (o + 3) & ~(size_t)3;
    L.wsum = o; o += (((size_t)TE * S + 63) / 64 + 1) * 4;
    L.queue = sense_queue_offset(L.wsum, TE, (int)S);
    if (!packed) o = L.queue + 8 + ((sense_max_items(TE, C, P, NV) + 3) & ~(size_t)3);
    L.total = (o + 15) & ~(size_t)15;
    return L;
}

enum ObstRec { OB_MX = 0, OB_MY, OB_RVX, OB_RVY, OB_DIST, OB_COUNT = 5 };  // per obstacle
enum PolyRec { PB_X = 0, PB_Y, PB_HEADING, PB_COUNT = 3 };                  // per polygon

// Running (inclusive) sum over the 64 lanes of a wave, every lane active:
// shifts within rows of 16 lanes, then the two row broadcasts.
__device__ __forceinline__ int wave_inclusive_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// One (primitive, ray) pair: build the ray, run the exact test.  p = primitive
// (circles first), el = env of the tile, r = ray, oid = obstacle id; `slot` =
// accumulator index of the ray.
struct PairHit {
    bool ok;
    int slot;
    unsigned int oid;
    unsigned long long key;
};

// A hit remembered between the two sweeps, in three registers: the key, and ray slot and obstacle id in one word,
// ~0 for none.  The id takes the low 4 bits in the compile-time shape (16 obstacles), else the low 12: a tile's LDS
// image ends below 4,096 obstacles (40 B each) and 2^15 ray slots (12 B each), so the two always fit.
template <bool FIXED>
struct KeptHit {
    static constexpr int kIdBits = FIXED ? 4 : 12;
    unsigned long long key;
    unsigned int a;
    __device__ __forceinline__ void clear() { a = 0xffffffffu; }
    __device__ __forceinline__ void set(const PairHit& h) {
        key = h.key;
        a = h.ok ? ((unsigned int)h.slot << kIdBits) | h.oid : 0xffffffffu;
    }
    __device__ __forceinline__ bool ok() const { return a != 0xffffffffu; }
    __device__ __forceinline__ int slot() const { return (int)(a >> kIdBits); }
    __device__ __forceinline__ unsigned int oid() const { return a & ((1u << kIdBits) - 1u); }
};

template <bool PACKED>
__device__ __forceinline__ PairHit pair_test(int p, int el, int r, unsigned int oid, int R, int n_circ, double range,
                                             const double* env, const double* circ, const double* seg,
                                             const double* ray_dirs) {
    const double cpsi = env[el * 2], spsi = env[el * 2 + 1];
    const double cb = ray_dirs[2 * r], sb = ray_dirs[2 * r + 1];
    const double dx = cpsi * cb - spsi * sb;
    const double dy = spsi * cb + cpsi * sb;
    double t = 0.0;
    bool ok;
    if (p < n_circ) {
        ok = circle_hit(circ[4 * p], circ[4 * p + 1], circ[4 * p + 2], dx, dy, t);
    } else {
        const double* sg = seg + (size_t)(p - n_circ) * (PACKED ? 4 : 5);
        // packed edges recompute a x e, with the operations phase 1 files it with
        ok = edge_hit(sg[0], sg[1], sg[2], sg[3], PACKED ? sg[0] * sg[3] - sg[1] * sg[2] : sg[4], dx, dy, t);
    }
    PairHit h;
    h.ok = ok && t < range;
    h.slot = el * R + r;
    h.oid = oid;
    h.key = t_key(t);
    return h;
}

// default risk index of one obstacle (m = centre relative to the vessel, rv =
// relative velocity, dist = signed hull distance)
__device__ __forceinline__ double risk_index(double mx, double my, double rvx, double rvy, double vv, double dist,
                                             double range, double inv_range, double w_d, double w_c, double t_c) {
    const double d = dist < 0.0 ? 0.0 : dist;
    const bool inrange = d < range;
    const double fd = 1.0 - div_by_const(inrange ? d : 0.0, range, inv_range);
    const double term_d = w_d * (fd * fd);
    const double closing = 0.0 - (mx * rvx + my * rvy);
    const bool has_c = (vv > 0.0) && (closing > 0.0);
    const double tcpa = (has_c ? closing : 0.0) / (has_c ? vv : 1.0);
    const double qx = mx + rvx * tcpa;
    const double qy = my + rvy * tcpa;
    const double dcpa = __builtin_sqrt(qx * qx + qy * qy);
    const double fc = 1.0 - div_by_const(dcpa < range ? dcpa : range, range, inv_range);
    const double term_c = (w_c * (fc * fc)) * (t_c / (t_c + tcpa));
    const double rr = term_d + (has_c ? term_c : 0.0);
    return inrange ? rr : 0.0;
}

// per-env outputs that need the scene: nearest obstacle, collision, reward, done
__device__ __forceinline__ void env_summary(const SenseArgs& a, int e, double bd, int bi) {
    const uint8_t coll = bd < a.contact_dist ? 1 : 0;
    const uint8_t goal = a.reached_goal[e];
    a.min_dist[e] = bd;
    a.min_idx[e] = bi;
    a.collision[e] = coll;
    double rew = 0.0;
    if (a.is_step) {
        const double* er = a.envrec + (size_t)e * ER_COUNT;
        rew = er[ER_REWARD];
        rew = rew - (coll ? a.rw_collision : 0.0);
        rew = rew + (goal ? a.rw_goal : 0.0);
        if (er[ER_RESET] != 0.0) rew = 0.0;  // first observation of a new episode
    }
    a.reward[e] = rew;
    a.done[e] = coll | goal | a.truncated[e];
}

// Diagnostic builds (-DAUV_PHASE_STAMPS) record the shader clock of every wave
// at the phase boundaries of k_sense; the product build compiles these away.
#ifdef AUV_PHASE_STAMPS
__device__ long long* g_stamps = nullptr;
#if AUV_PHASE_STAMPS == 2   // only the two device-wide stamps: the kernel's registers and timing stay what they are
#define AUV_STAMP(n)
#else
#define AUV_STAMP(n)                                                                                  \
    do {                                                                                              \
        if (g_stamps && (threadIdx.x & 63) == 0)                                                      \
            g_stamps[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (n)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#endif
// the device-wide 100 MHz counter (the shader clocks of different CUs cannot be compared with one another): slots
// 14 and 15 of a wave's record, at the first and the last stamp
#define AUV_STAMP_REAL(n)                                                                             \
    do {                                                                                              \
        if (g_stamps && (threadIdx.x & 63) == 0)                                                      \
            g_stamps[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (n)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define AUV_STAMP(n)
#define AUV_STAMP_REAL(n)
#endif

// Debug builds (-DAUV_DEBUG_BOUNDS) count every violated index bound of the
// on-chip arrays instead of trusting the layout arithmetic; the product build
// compiles the checks away.
#ifdef AUV_DEBUG_BOUNDS
__device__ unsigned long long g_violations = 0ull;
#define AUV_CHECK(cond)                                  \
    do {                                                 \
        if (!(cond)) atomicAdd(&g_violations, 1ull);     \
    } while (0)
#else
#define AUV_CHECK(cond)
#endif


// The scene shape of the reference's default scenario at the benchmark's size:
// 180 rays (utils/dataloader.py:13), 8 circles + 8 five-vertex ships per env,
// tiles of 4 envs on 256 threads.  k_sense has an instantiation with these as
// compile-time constants (index arithmetic folds, loops over vertices unroll,
// a dozen scalar registers are freed); every other shape takes the generic one.
struct DefaultShape {
    static constexpr int R = 180, C = 8, P = 8, NV = 5, TE = 4, NTHR = 256;
};
inline bool is_default_shape(const Dims& d, int te, int block) {
    return d.R == DefaultShape::R && d.C == DefaultShape::C && d.P == DefaultShape::P && d.NV == DefaultShape::NV &&
           te == DefaultShape::TE && block == DefaultShape::NTHR;
}

// RISK: 0 = default risk index, 1 = the reference's index (auv_risk.hpp)
// FIXED: the shape is DefaultShape (checked by the launcher)
typedef const __attribute__((address_space(4))) SenseArgs* KernArgs;
// The kernel's arguments re-read from the kernarg segment at the point of the call: values fetched here are not
// held in scalar registers (or spilled to vector lanes) from the launch on.
__device__ __forceinline__ SenseArgs reload_args() {
    KernArgs p = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    SenseArgs r;
    __builtin_memcpy(&r, (const SenseArgs*)p, sizeof(SenseArgs));
    return r;
}

// QUEUED (general shape only): the work of the interval between the first two barriers is drawn from a counter in
// LDS instead of being dealt by wave number (see there); the launcher picks it for tiles whose work is long and
// uneven.
template <int RISK, bool FIXED, bool QUEUED = false>
__device__ __forceinline__ void sense_tile(const SenseArgs& a0) {
    static_assert(!(FIXED && QUEUED), "the compile-time shape is dealt by wave number");
    const SenseArgs& a = a0;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int V = a.d.V;
    const int R = FIXED ? DefaultShape::R : a.d.R, C = FIXED ? DefaultShape::C : a.d.C, P = FIXED ? DefaultShape::P : a.d.P;
    const int NV = FIXED ? DefaultShape::NV : a.d.NV, O = FIXED ? DefaultShape::C + DefaultShape::P : a.d.O;
    const int S = P * NV;
    const int TE = FIXED ? DefaultShape::TE : a.TE;
    struct { int env, circ, seg, best, obst, edist, ids, ecross, live, wsum; } L = {
        a.lds[0], a.lds[1], a.lds[2], a.lds[3], a.lds[4], a.lds[5], a.lds[6], a.lds[7], a.lds[8], a.lds[9]};
    if (FIXED) {
        // constants but for the room an attached network takes in front of best[]
        const SenseLayout T = sense_layout(TE, R, C, P, NV, a.cnn_w ? a.cnn_kind : -1, true);
        L = {(int)T.env, (int)T.circ, (int)T.seg, (int)T.best, (int)T.obst, (int)T.edist, (int)T.ids, (int)T.ecross,
             (int)T.live, (int)T.wsum};
    }
    double* env = reinterpret_cast<double*>(lds_raw + L.env);
    double* circ = reinterpret_cast<double*>(lds_raw + L.circ);
    double* seg = reinterpret_cast<double*>(lds_raw + L.seg);
    unsigned long long* best = reinterpret_cast<unsigned long long*>(lds_raw + L.best);
    double* obst = reinterpret_cast<double*>(lds_raw + L.obst);
    const int ob_n = TE * O;  // obstacle slots of the tile = stride between the fields of obst[]
    double* pobs = obst + (size_t)OB_COUNT * ob_n;  // polygon-only fields
    const int pb_n = TE * P;
    double* edist = reinterpret_cast<double*>(lds_raw + L.edist);
    unsigned int* ids = reinterpret_cast<unsigned int*>(lds_raw + L.ids);
    int* ecross = reinterpret_cast<int*>(lds_raw + L.ecross);
    // crossing records and edge numbers: a word and a half-word each, or (compile-time shape) a byte
    auto ecross_put = [&](int idx, int v) {
        if (FIXED) reinterpret_cast<unsigned char*>(ecross)[idx] = (unsigned char)v;
        else ecross[idx] = v;
    };
    auto ecross_get = [&](int idx) -> int {
        return FIXED ? (int)reinterpret_cast<const unsigned char*>(ecross)[idx] : ecross[idx];
    };
    // numbers of the edges that have candidate rays, in the order their waves found them
    auto live_put = [&](int idx, int edge) {
        if (FIXED) (lds_raw + L.live)[idx] = (unsigned char)edge;
        else reinterpret_cast<unsigned short*>(lds_raw + L.live)[idx] = (unsigned short)edge;
    };
    auto live_get = [&](int idx) -> int {
        return FIXED ? (int)(lds_raw + L.live)[idx] : (int)reinterpret_cast<const unsigned short*>(lds_raw + L.live)[idx];
    };
    int* wsum = reinterpret_cast<int*>(lds_raw + L.wsum);
    // (queued tiles) two counters, then a byte per work item
    int* queue = reinterpret_cast<int*>(lds_raw + (QUEUED ? (int)sense_queue_offset((size_t)L.wsum, TE, S) : 0));
    unsigned char* item_flag = reinterpret_cast<unsigned char*>(queue) + 8;
    const int e0 = blockIdx.x * TE;
    const int ne = min(TE, V - e0);
    const int n_circ = ne * C;
    const int tid = threadIdx.x;
    const int nthr = FIXED ? DefaultShape::NTHR : (int)blockDim.x;
    const int lane = tid & 63;
    const int n_waves = nthr >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double range = a.sensor_range;
    const float frange = (float)range * kRangeSlack;
    const float inv_delta = (float)R * (1.0f / 6.2831853f);
    const float half_scan = 0.5f * (float)R;
    const float margin = fmaxf(kIvMargin, kIvMarginRad * inv_delta);
    // reciprocals for fast_div: the 1-ulp hardware reciprocal is well within its error budget
    const float inv_C = __builtin_amdgcn_rcpf((float)(C > 0 ? C : 1)), inv_S = __builtin_amdgcn_rcpf((float)(S > 0 ? S : 1));
    const float inv_NV = __builtin_amdgcn_rcpf((float)NV), inv_O = __builtin_amdgcn_rcpf((float)(O > 0 ? O : 1));

    AUV_STAMP(0);
    AUV_STAMP_REAL(14);
    // ---- phase 0: accumulators (first read after the barrier that ends phase 1)
    // heading (and its sin/cos) of the tile's vessels: requested now, filed at the
    // end of phase 1, so that the first barrier does not wait for global memory
    double env_c = 0.0, env_s = 0.0, env_psi = 0.0;
    if (tid < ne) {  // ne <= 64 <= nthr
        const double* er = a.envrec + (size_t)(e0 + tid) * ER_COUNT;
        env_c = er[ER_COS];
        env_s = er[ER_SIN];
        env_psi = a.state[2 * (size_t)V + e0 + tid];
    }
    if (QUEUED && tid == 0) {
        // the first sweep items are dealt by wave number (sweep 0: to the waves without a job; sweep 1: to all):
        // the draws start behind those
        // (used by the queued tiles only; their number of job waves as computed below)
        const int rounds = max(1, (ne * O + 63) >> 6), tasks = a.with_meta ? 2 : 1, nw = nthr >> 6;
        queue[0] = nw - min(tasks * rounds, max(tasks, (2 * nw) / 3));
        queue[1] = nw;
    }
    // The per-ray arrays are walked four rays per lane when the ray count allows (every tile then starts at a
    // multiple of four rays, in LDS and in memory alike): 16-byte LDS accesses and stores, a quarter of the loop and
    // address instructions, which are most of what these loops execute.
    const bool by_four = FIXED || (R & 3) == 0;
    const unsigned long long range_bits = (unsigned long long)__double_as_longlong(range);
    if (by_four) {
        const auv_u64x2 rb2 = {range_bits, range_bits};
        const auv_u32x4 none4 = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        for (int q = tid; q < (ne * R) >> 2; q += nthr) {
            reinterpret_cast<auv_u64x2*>(best)[2 * q] = rb2;
            reinterpret_cast<auv_u64x2*>(best)[2 * q + 1] = rb2;
            reinterpret_cast<auv_u32x4*>(ids)[q] = none4;
        }
    } else {
        for (int f = tid; f < ne * R; f += nthr) {
            best[f] = range_bits;
            ids[f] = 0xffffffffu;
        }
    }
    AUV_STAMP(1);

    // ---- phase 1: geometry of the primitives.  Circles go to the highest
    // thread ids, so that on small tiles they do not share a wave with the
    // first edges.
    const int wave0 = wave << 6;  // first thread of this wave
    for (int ib = 0; ib < n_circ; ib += nthr) {
        if (ib + nthr - 64 - wave0 >= n_circ) continue;  // nothing for this wave
        const int i = ib + nthr - 1 - tid;
        if (i < n_circ) {
            const int el = fast_div(i, inv_C);
            const int j = i - el * C;
            const int e = e0 + el;
            double* c = a.circles + ((size_t)e0 * C + i) * 5;
            double cx = c[0], cy = c[1];
            const double rad = c[2], vx = c[3], vy = c[4];
            const double* er = a.envrec + (size_t)e * ER_COUNT;
            if (a.is_step && er[ER_RESET] == 0.0) {  // a restarted env's obstacles were just put back by k_vessel
                cx = cx + vx * a.dt;
                cy = cy + vy * a.dt;
                c[0] = cx;
                c[1] = cy;
            }
            const double mx = cx - a.state[e];
            const double my = cy - a.state[(size_t)V + e];
            const double mm = mx * mx + my * my;
            const double cc = mm - rad * rad;
            const bool on = rad > 0.0;
            circ[i * 4 + 0] = on ? mx : 0.0;
            circ[i * 4 + 1] = on ? my : 0.0;
            circ[i * 4 + 2] = on ? cc : __builtin_inf();
            circ[i * 4 + 3] = on ? rad : 0.0;
            // obstacle record for phase 2
            double* ob = obst + (el * O + j);
            ob[OB_MX * ob_n] = mx;
            ob[OB_MY * ob_n] = my;
            ob[OB_RVX * ob_n] = vx - er[ER_VOX];
            ob[OB_RVY * ob_n] = vy - er[ER_VOY];
            ob[OB_DIST * ob_n] = on ? (__builtin_sqrt(mm) - rad) - a.vessel_radius : __builtin_inf();
            AUV_CHECK(i >= 0 && i < TE * C && el >= 0 && el < ne && j >= 0 && j < C);
        }
    }
    // Edges: geometry, distance and crossing bit of every edge; the few that a
    // ray can hit at all (in range, facing the vessel) are listed, and get their
    // candidate rays after the barrier from one wave instead of occupying a lane
    // in every wave here.  Every block of 64 edges has a list and a count of its
    // own: no counter is shared, so nothing has to be zeroed behind a barrier.
    for (int ib = 0; ib < ne * S; ib += nthr) {
        if (ib + wave0 >= ne * S) continue;  // nothing for this wave
        const int i = ib + tid;
        bool has_rays = false;
        if (i < ne * S) {
            const int el = fast_div(i, inv_S);
            const int s = i - el * S;
            const int q = fast_div(s, inv_NV);
            const int k = s - q * NV;
            const int e = e0 + el;
            const size_t ep = (size_t)e * P + q;
            const int nv = a.poly_nv[ep];
            const double reach2 = a.poly_reach2[ep];
            const double* p = a.poly_pose + ep * 6;
            // every vertex slot exists in memory, so the loads need not wait for nv
            const double* vb = a.poly_rot + ep * NV * 2;  // already turned to the ship's heading
            const int k1c = (k + 1 < NV) ? k + 1 : 0;
            const double vkx = vb[2 * k], vky = vb[2 * k + 1];
            const double v1x = vb[2 * k1c], v1y = vb[2 * k1c + 1], v0x = vb[0], v0y = vb[1];
            double X = p[0], Y = p[1];
            const double pvx = p[4], pvy = p[5];
            if (a.is_step && a.envrec[(size_t)e * ER_COUNT + ER_RESET] == 0.0) {
                X = X + pvx * a.dt;
                Y = Y + pvy * a.dt;
            }
            const double px = a.state[e], py = a.state[(size_t)V + e];
            if (k == 0) {
                // one edge thread per polygon files its obstacle record for phase 2
                const double* er = a.envrec + (size_t)e * ER_COUNT;
                double* ob = obst + (el * O + C + q);
                ob[OB_MX * ob_n] = X - px;
                ob[OB_MY * ob_n] = Y - py;
                ob[OB_RVX * ob_n] = pvx - er[ER_VOX];
                ob[OB_RVY * ob_n] = pvy - er[ER_VOY];
                double* pb = pobs + (el * P + q);
                pb[PB_X * pb_n] = X;  // advanced position, written back in phase 2
                pb[PB_Y * pb_n] = Y;
                // the heading column of the CPA index, or (RISK == 1) the ship's beam for the reference's distance
                pb[PB_HEADING * pb_n] = RISK == 1 ? a.poly_beam[ep] : a.poly_heading[ep];
            }
            double ax = 0.0, ay = 0.0, ex = 0.0, ey = 0.0, nt = 0.0, dd = __builtin_inf();
            int cross = 0;
            if (k < nv) {
                const bool wrap = k + 1 >= nv;  // the last edge closes on vertex 0
                const double bx = wrap ? v0x : v1x, by_ = wrap ? v0y : v1y;
                const double Ax = X + vkx;
                const double Ay = Y + vky;
                const double Bx = X + bx;
                const double By = Y + by_;
                ax = Ax - px;
                ay = Ay - py;
                const double by = By - py;
                ex = Bx - Ax;
                ey = By - Ay;
                nt = ax * ey - ay * ex;
                // point-to-edge distance and crossing-number bit (vessel at the origin)
                const double ee = ex * ex + ey * ey;
                double tp = 0.0;
                if (ee > 0.0) {
                    tp = (0.0 - (ax * ex + ay * ey)) / ee;
                    tp = tp < 0.0 ? 0.0 : tp;
                    tp = tp > 1.0 ? 1.0 : tp;
                }
                const double qx = ax + tp * ex;
                const double qy = ay + tp * ey;
                dd = qx * qx + qy * qy;  // squared: the one square root per polygon is taken in phase 2
                const bool up = (ay <= 0.0) && (by > 0.0) && (nt > 0.0);
                const bool dn = (by <= 0.0) && (ay > 0.0) && (nt < 0.0);
                cross = (up || dn) ? 1 : 0;
                // an edge of a convex polygon that faces away from a vessel outside it is no LiDAR surface
                // (polygon_reach2); nor is one wholly beyond sensor range
                const double cmx = X - px, cmy = Y - py;
                const bool faces_away = reach2 >= 0.0 && (cmx * cmx + cmy * cmy) > reach2 && nt > 0.0;
                has_rays = !(fast_sqrtf((float)dd) > frange || faces_away);
            }
            double* sg = seg + (size_t)i * (FIXED ? 4 : 5);
            sg[0] = ax;
            sg[1] = ay;
            sg[2] = ex;
            sg[3] = ey;
            if (!FIXED) sg[4] = nt;
            edist[i] = dd;
            ecross_put(i, k == 0 ? (cross | (nv << 1)) : cross);  // edge 0 also carries the vertex count
            AUV_CHECK(i >= 0 && i < TE * S && el >= 0 && el < ne && q >= 0 && q < P);
        }
        const unsigned long long found = __ballot(has_rays);
        const int at = ib + wave0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(found >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((unsigned int)found, 0u));
        AUV_CHECK(!has_rays || (at >= 0 && at < TE * S));
        if (has_rays) live_put(at, i);
        if (lane == 0) wsum[(ib + wave0) >> 6] = __popcll(found);
    }
    if (tid < ne) {
        env[tid * 2] = env_c;
        env[tid * 2 + 1] = env_s;
        env[2 * TE + tid] = env_psi;
    }
    AUV_STAMP(2);
    __syncthreads();
    AUV_STAMP(3);

    // ---- phase 2 and 3 share a barrier interval.  Phase 2, the obstacle-level
    // outputs, is two tasks, each on a wave of its own; the other waves (all of
    // them, when the workgroup has no more than that) run phase 3, the LiDAR
    // sweep, in units of 64 primitives: the circles in thread order, then the
    // listed edges.  A unit's wave gives each primitive its candidate rays
    // [first, first + cnt), numbers the unit's (primitive, ray) pairs 0..total-1
    // in lane order and shares them out evenly: pair j goes to lane j % 64, which
    // finds the primitive by bisection over the running counts.  So the time goes
    // with the number of pairs, not with the widest primitive.
    //
    // Work items of the interval.  In the compile-time shape the tasks of phase 2
    // have the first waves and the sweep's units the others.  In the general
    // shape the jobs of phase 2 (task x round of 64 obstacles) have the first
    // waves, as many as the obstacle rounds call for, and the sweep's (unit,
    // part) pairs are drawn from a counter in LDS by the waves as they become
    // free, the other waves' first items being their own numbers: a dense
    // scene's obstacle rounds and a long unit's rounds spread over the whole
    // workgroup whatever their sizes.
    const int n_tasks = a.with_meta ? 2 : 1;
    const int n_rounds = FIXED ? 1 : max(1, (ne * O + 63) >> 6);
    const int n_jobs = n_tasks * n_rounds;
    const int n_blocks = (ne * S + 63) >> 6;
    int n_live = 0;
    for (int b = 0; b < n_blocks; ++b) n_live += wsum[b];
    const int units_c = (n_circ + 63) >> 6, n_units = units_c + ((n_live + 63) >> 6);
    // Tiles whose work is long and uneven -- three rounds of obstacles or more, or twice the rays a unit's rounds
    // are sized for (sense_wants_queue) -- take it `queued`: the jobs on two thirds of the waves, the sweep items
    // drawn from the counter.
    // The others keep the fixed deal: the two tasks on the first two waves (all the waves, when there are no
    // more), a task's rounds in turn on its wave, the units' parts round the other waves; what the counter and
    // its flags cost, and what a sweep item that is not its wave's first costs in sweep 1, outweigh the balance.
    const int by_rays = (R + 179) / 180;
    constexpr bool queued = QUEUED;
    const int n_job_waves = queued ? min(n_jobs, max(n_tasks, (2 * n_waves) / 3)) : (n_waves > n_tasks ? n_tasks : n_waves);
    const int free_waves = n_waves > n_tasks || queued ? n_waves - n_job_waves : 0;
    // A unit's rounds are dealt into n_share parts, each an item of its own (its wave opens the unit again): more
    // parts the more rays a primitive can cover, and with waves to spare enough for each to find one.
    const int by_waves = n_units == 0 ? 1 : free_waves >= 2 * n_units ? free_waves / n_units
                         : (free_waves == 0 && n_waves >= 2 * n_units) ? n_waves / n_units : 1;
    const int n_share = (FIXED || n_units == 0) ? 1 : QUEUED ? min(kShareMax, max(by_rays, by_waves)) : by_waves;
    const int n_sweep = n_units * n_share;
    AUV_CHECK(!QUEUED || n_jobs + n_sweep <= (int)sense_max_items(TE, C, P, NV));
    // the next item off counter `which`, the same in every lane; items >= n_items: the list has run out
    auto draw = [&](int which, int base) -> int {
        int it = 0;
        if (lane == 0) it = atomicAdd(&queue[which], 1);
        return base + __builtin_amdgcn_readfirstlane(it);
    };
    // the fixed deal: the waves that sweep
    const int first_ray_wave = FIXED ? n_tasks : (n_waves > n_tasks ? n_tasks : 0), n_ray_waves = n_waves - first_ray_wave;
    struct Unit {
        int incl, excl, total;      // running count of pairs up to and with this lane's primitive, without it; all
        unsigned int desc, desc2;   // first ray, primitive, env and obstacle id of the lane's primitive
    };
    auto unit_open = [&](int u) -> Unit {
        int2 iv = make_int2(0, 0);
        int prim = 0, el = 0, oid = 0;
        if (u < units_c) {
            const int i = (u << 6) + lane;
            if (i < n_circ) {
                el = fast_div(i, inv_C);
                oid = i - el * C;
                prim = i;
                const double* c4 = circ + 4 * i;
                const float fr = (float)c4[3], fmx = (float)c4[0], fmy = (float)c4[1];
                const float fmm = __builtin_fmaf(fmx, fmx, fmy * fmy);
                if (!(fr > 0.0f)) {
                    // absent slot
                } else if (!(fmm > fr * fr * 1.001f)) {
                    iv.y = R;  // vessel inside or touching: every ray can hit
                } else if (fast_sqrtf(fmm) - fr > frange) {
                    // wholly beyond sensor range
                } else {
                    const float xc = scan_index(fast_atan2f(fmy, fmx), (float)env[2 * TE + el], inv_delta);
                    const float hw = fast_atan2f(fr, fast_sqrtf(__builtin_fmaf(-fr, fr, fmm))) * inv_delta;
                    iv = make_interval(xc - hw, xc + hw, R, margin);
                }
            }
        } else {
            const int j = ((u - units_c) << 6) + lane;
            // candidate rays of listed edge i: the arc between the bearings of its two endpoints
            auto edge_interval = [&](int i) {
                AUV_CHECK(i >= 0 && i < ne * S);
                el = fast_div(i, inv_S);
                oid = C + fast_div(i - el * S, inv_NV);
                prim = n_circ + i;
                const double* sg = seg + (size_t)i * (FIXED ? 4 : 5);
                const float fax = (float)sg[0], fay = (float)sg[1], fex = (float)sg[2], fey = (float)sg[3];
                const float fbx = fax + fex, fby = fay + fey;
                const float dmin = fast_sqrtf((float)edist[i]);
                const float aa = __builtin_fmaf(fax, fax, fay * fay), bb = __builtin_fmaf(fbx, fbx, fby * fby);
                if (aa < 1e-6f || bb < 1e-6f || dmin < 1e-3f) {
                    iv.y = R;  // an endpoint (or the edge) is on top of the vessel: bearings are meaningless
                } else {
                    const float fpsi = (float)env[2 * TE + el];
                    const float xa = scan_index(fast_atan2f(fay, fax), fpsi, inv_delta);
                    const float xb = scan_index(fast_atan2f(fby, fbx), fpsi, inv_delta);
                    float d = xb - xa;
                    d = d > half_scan ? d - (float)R : d;
                    d = d < -half_scan ? d + (float)R : d;
                    if (fabsf(d) > half_scan - 1.5f) {
                        iv.y = R;  // subtends ~180 deg: the short way round is ambiguous
                    } else {
                        const float lo = d >= 0.0f ? xa : xa + d;
                        iv = make_interval(lo, lo + fabsf(d), R, margin);
                    }
                }
            };
            if constexpr (!QUEUED) {
                if (j < n_live) {
                    // the j-th listed edge: walk the blocks' counts
                    int b = 0, before = 0;
                    for (int nb = wsum[0]; j >= before + nb; nb = wsum[b]) {
                        before += nb;
                        ++b;
                    }
                    AUV_CHECK(b >= 0 && b < n_blocks);
                    edge_interval(live_get((b << 6) + (j - before)));
                }
            } else {
                // Many blocks: their counts (64 blocks at most: the LDS image ends at 4,096 edges) summed over the
                // wave once, every lane taking part; a lane then finds its block by bisection instead of walking
                // the counts one LDS read after the other.
                AUV_CHECK(n_blocks <= 64);
                const int blk_cnt = lane < n_blocks ? wsum[lane] : 0;
                const int blk_incl = wave_inclusive_sum(blk_cnt);
                int lo4 = 0;  // 4 x the number of blocks whose listed edges all come before the j-th
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const int probe = __builtin_amdgcn_ds_bpermute(lo4 + (step - 1) * 4, blk_incl);
                    lo4 = probe <= j ? lo4 + step * 4 : lo4;
                }
                const int before = __builtin_amdgcn_ds_bpermute(lo4, blk_incl - blk_cnt);
                if (j < n_live) {
                    AUV_CHECK((lo4 >> 2) < n_blocks && j - before >= 0 && j - before < 64);
                    edge_interval(live_get(((lo4 >> 2) << 6) + (j - before)));
                }
            }
        }
        AUV_CHECK(iv.x >= 0 && iv.x < R && iv.y >= 0 && iv.y <= R && el >= 0 && el < ne && oid >= 0 && oid < O);
        Unit U;
        U.incl = wave_inclusive_sum(iv.y);
        U.excl = U.incl - iv.y;
        U.total = __builtin_amdgcn_readlane(U.incl, 63);
        // the compile-time shape fits the four in a word: 180 rays, 4 envs, 16 obstacles, 192 primitives
        U.desc = FIXED ? (unsigned)iv.x | ((unsigned)el << 8) | ((unsigned)oid << 10) | ((unsigned)prim << 14)
                       : (unsigned)iv.x | ((unsigned)prim << 16);
        U.desc2 = (unsigned)el | ((unsigned)oid << 8);
        return U;
    };
    auto unit_pair = [&](const Unit& U, int j) -> PairHit {
        int lo4 = 0;  // 4 x the number of lanes whose pairs all come before pair j
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const int probe = __builtin_amdgcn_ds_bpermute(lo4 + (step - 1) * 4, U.incl);
            lo4 = probe <= j ? lo4 + step * 4 : lo4;
        }
        const int k = j - __builtin_amdgcn_ds_bpermute(lo4, U.excl);
        const unsigned int d = (unsigned int)__builtin_amdgcn_ds_bpermute(lo4, (int)U.desc);
        int first, prim, el;
        unsigned int oid;
        if (FIXED) {
            first = (int)(d & 0xffu);
            el = (int)((d >> 8) & 3u);
            oid = (d >> 10) & 15u;
            prim = (int)(d >> 14);
        } else {
            const unsigned int d2 = (unsigned int)__builtin_amdgcn_ds_bpermute(lo4, (int)U.desc2);
            first = (int)(d & 0xffffu);
            prim = (int)(d >> 16);
            el = (int)(d2 & 0xffu);
            oid = d2 >> 8;
        }
        int r = first + k;
        r = r >= R ? r - R : r;
        PairHit h;
        h.ok = false;
        h.slot = 0;
        h.oid = 0u;
        h.key = 0ull;
        if (j < U.total) {
            AUV_CHECK(k >= 0 && r >= 0 && r < R && prim >= 0 && prim < n_circ + ne * S && el >= 0 && el < ne);
            h = pair_test<FIXED>(prim, el, r, oid, R, n_circ, range, env, circ, seg, a.ray_dirs);
            AUV_CHECK(!h.ok || (h.slot >= 0 && h.slot < ne * R && (int)h.oid < O));
        }
        return h;
    };
    {
        // The kernel's arguments are re-read where each task uses them (reload_args), a task's own few at a time:
        // fetched once at the head of the interval they would all stay live across it, most of them spilled.
        const int n_obst = ne * O;
        // signed hull distance less the vessel radius; polygons: nearest edge
        // (sqrt of the smallest squared distance), sign from the crossing parity
        auto hull_distance = [&](const SenseArgs& a, const double* ob, int el, int o) -> double {
            if (o < C) return ob[OB_DIST * ob_n];
            const int s0 = el * S + (o - C) * NV;
            const int nv = ecross_get(s0) >> 1;
            AUV_CHECK(s0 >= 0 && s0 + NV <= TE * S && nv >= 0 && nv <= NV);
            double dmin2 = __builtin_inf();
            int par = 0;
            for (int k = 0; k < nv; ++k) {
                const double d2 = edist[s0 + k];
                dmin2 = d2 < dmin2 ? d2 : dmin2;
                par ^= ecross_get(s0 + k) & 1;
            }
            const double dmin = __builtin_sqrt(dmin2);
            return nv > 0 ? (par ? -dmin : dmin) - a.vessel_radius : __builtin_inf();
        };
        // the dist column and risk input of the reference's index: centre distance
        // less size less clearance; a ship's size is its beam (body-frame vertices
        // across its heading), a circle's its radius, i.e. the hull distance
        auto ref_distance = [&](const SenseArgs& a, const double* ob, int el, int o, double mx, double my) -> double {
            if (o < C) return ob[OB_DIST * ob_n];
            const int nv = ecross_get(el * S + (o - C) * NV) >> 1;
            const double beam = pobs[PB_HEADING * pb_n + (el * P + (o - C))];
            return nv > 0 ? (__builtin_sqrt(mx * mx + my * my) - beam) - a.vessel_radius : __builtin_inf();
        };
        // when an env's obstacles sit in an aligned power-of-two group of lanes
        // of one wave, its nearest obstacle is a shuffle reduction
        const bool lanes_per_env = O > 0 && (O & (O - 1)) == 0 && TE * O <= 64;
        auto run_job = [&](int task, int i0) {
            {
                const int i = i0 + lane;
                const bool live = i < n_obst;
                const int el = live ? fast_div(i, inv_O) : 0;
                const int o = live ? i - el * O : 0;
                const int e = e0 + el;
                double dist = __builtin_inf();
                if (live) {
                    double* ob = obst + i;
                    const double mx = ob[OB_MX * ob_n], my = ob[OB_MY * ob_n], rvx = ob[OB_RVX * ob_n], rvy = ob[OB_RVY * ob_n];
                    const size_t eo = (size_t)e * O + o;
                    if (task == 0) {
                        const SenseArgs a = reload_args();
                        double* m = a.meta + eo * 5;
                        dist = hull_distance(a, ob, el, o);
                        double hd = 0.0;
                        if (o >= C) {
                            const double* pb = pobs + (el * P + (o - C));
                            if (a.is_step) {
                                // every edge thread has read the old position before the barrier
                                double* pp = a.poly_pose + ((size_t)e * P + (o - C)) * 6;
                                pp[0] = pb[PB_X * pb_n];
                                pp[1] = pb[PB_Y * pb_n];
                            }
                            hd = pb[PB_HEADING * pb_n];
                        }
                        a.obst_dist[eo] = dist;
                        m[1] = RISK == 1 ? ref_distance(a, ob, el, o, mx, my) : dist;
                        if (RISK == 0) {
                            m[0] = hd;
                            m[2] = __builtin_sqrt(rvx * rvx + rvy * rvy);
                        }
                    } else {
                        // task 1: bearing, course, relative speed and risk of the obstacle, on one wave: the
                        // bearing and the reference's relative motion are formed once and shared
                        const SenseArgs a = reload_args();
                        double* m = a.meta + eo * 5;
                        // RISK == 1: the heading, speed and course columns follow the reference's conventions
                        const double* ov = o < C ? a.circles + ((size_t)e * C + o) * 5 + 3
                                                 : a.poly_pose + ((size_t)e * P + (o - C)) * 6 + 4;
                        const double psi = a.state[2 * (size_t)V + e];
                        const double bearing = wrap_angle(atan2_det(my, mx) - psi);
                        m[3] = bearing;
                        if (RISK == 1) {
                            const double* er = a.envrec + (size_t)e * ER_COUNT;
                            const RefMotion rm = reference_motion(ov[0], ov[1], er[ER_VOX], er[ER_VOY], er[ER_COS], er[ER_SIN], psi);
                            m[2] = rm.still ? rm.ss : __builtin_sqrt(rm.rr);
                            m[4] = rm.still ? (psi > 0.0 ? 0.0 : 0.0 - kPi) : atan2_det(rm.wy, rm.wx);
                            if (a.refresh_heading || er[ER_RESET] != 0.0)
                                m[0] = (ov[0] == 0.0 && ov[1] == 0.0) ? 0.0 : atan2_det(ov[1], ov[0]);
                            const double dref = ref_distance(a, ob, el, o, mx, my);
                            a.risk[eo] = cri_sum(risk_reference_s_of(rm, mx, my, dref, dref < __builtin_inf()),
                                                 risk_reference_t(mx, my, ov[0], ov[1], bearing));
                        } else {
                            m[4] = wrap_angle(atan2_det(rvy, rvx) - psi);
                            a.risk[eo] = risk_index(mx, my, rvx, rvy, rvx * rvx + rvy * rvy, hull_distance(a, ob, el, o), range,
                                                    a.inv_range, a.w_d, a.w_c, a.t_c);
                        }
                    }
                }
                if (task == 0 && lanes_per_env) {
                    // (dist, id) lexicographic minimum over the env's O lanes; all 64 lanes take part
                    double bd = dist;
                    int bi = o;
                    for (int d = O >> 1; d >= 1; d >>= 1) {
                        const double od = __shfl_xor(bd, d, 64);
                        const int oi = __shfl_xor(bi, d, 64);
                        if (od < bd || (od == bd && oi < bi)) {
                            bd = od;
                            bi = oi;
                        }
                    }
                    if (live && o == 0) env_summary(reload_args(), e, bd, bd < __builtin_inf() ? bi : -1);
                } else if (task == 0 && live && o >= C) {
                    // general shape: park the polygon's distance beside the
                    // circles' for the per-env minimum taken after the barrier
                    // (no other task reads this slot)
                    obst[OB_DIST * ob_n + i] = dist;
                }
            }
        };
        if constexpr (!QUEUED) {
            for (int task = wave; task < n_tasks; task += n_waves) {
                for (int i0 = 0; i0 < n_obst || (task == 0 && i0 == 0); i0 += 64) run_job(task, i0);
                if (task == 0 && !lanes_per_env) {
                    // the wave that parked the distances walks them per env
                    for (int el = lane; el < ne; el += 64) {
                        double bd = __builtin_inf();
                        int bi = -1;
                        for (int o = 0; o < O; ++o) {
                            const double d = obst[OB_DIST * ob_n + el * O + o];
                            if (d < bd) {
                                bd = d;
                                bi = o;
                            }
                        }
                        env_summary(reload_args(), e0 + el, bd, bi);
                    }
                }
            }
        } else {
            for (int job = wave; wave < n_job_waves && job < n_jobs; job += n_job_waves)
                run_job(job % n_tasks, (job / n_tasks) << 6);
        }
    }
    // (queued tiles: an env's obstacles may be on several waves, its nearest one is found behind the barrier)
    const bool summary_after_barrier = QUEUED && !(O > 0 && (O & (O - 1)) == 0 && TE * O <= 64);
    AUV_STAMP(4);
    // sweep 0: nearest hit of every ray.  The hits of the first rounds of the
    // wave's first unit stay in registers for sweep 1, which settles the
    // obstacle id among equal distances; whatever is beyond is walked again.
    constexpr int kKeep = FIXED ? 6 : 4;
    KeptHit<FIXED> kept[kKeep];
#pragma unroll
    for (int j = 0; j < kKeep; ++j) kept[j].clear();
    bool walk_again = false;  // same for the whole wave
    // returns: bit 0 = rounds of this part are left for sweep 1 to walk again, bit 1 = its first rounds were kept
    auto sweep0 = [&](int u, int part, bool& first_unit) -> int {
        const Unit U = unit_open(u);
        int j0 = part * 64;
        int flags = (QUEUED && first_unit) ? 2 : 0;
        if (first_unit) {
#pragma unroll
            for (int j = 0; j < kKeep; ++j) {
                if (j0 < U.total) {
                    const PairHit h = unit_pair(U, j0 + lane);
                    if (h.ok) atomicMin(&best[h.slot], h.key);
                    kept[j].set(h);
                }
                j0 += n_share * 64;
            }
            first_unit = false;
        }
        for (; j0 < U.total; j0 += n_share * 64) {
            const PairHit h = unit_pair(U, j0 + lane);
            if (h.ok) atomicMin(&best[h.slot], h.key);
            if (QUEUED) flags |= 1;
            else walk_again = true;
        }
        return flags;
    };
    if constexpr (!QUEUED) {
        if (wave >= first_ray_wave) {
            bool first_unit = true;
            for (int w = wave - first_ray_wave; w < n_units * n_share; w += n_ray_waves) {
                const int u = FIXED ? w : w / n_share, part = FIXED ? 0 : w - u * n_share;  // rounds part, part + n_share, ...
                sweep0(u, part, first_unit);
            }
        }
    } else {
        bool first_unit = true;
        // sweep items off the same counter; what each leaves for sweep 1 is filed per item, for whichever wave
        // draws it there
        for (int w = wave >= n_job_waves ? wave - n_job_waves : draw(0, 0); w < n_sweep; w = draw(0, 0)) {
            const int u = w / n_share;
            const int f = sweep0(u, w - u * n_share, first_unit);  // rounds part, part + n_share, ...
            if (lane == 0) item_flag[w] = (unsigned char)f;
        }
    }
    AUV_STAMP(5);
    __syncthreads();
    AUV_STAMP(6);
    // ---- phase 4: per-ray outputs, 16 B per ray, coalesced.  The distances
    // are final, so ranges and closeness go out now, beside sweep 1; with a
    // network attached its input and its weights are filed in the bytes of
    // obst[], which phase 2 is done with.
    const SenseArgs a4 = reload_args();
    const SenseArgs& a_out = a4;
    float* cnn_x = reinterpret_cast<float*>(lds_raw + L.obst);  // [TE][R] closeness, then the weights
    float* cnn_wl = cnn_x + TE * R;
    const int n_w = a_out.cnn_w ? cnn_n_weights(a_out.cnn_kind) : 0;
    float w_mine = 0.f;
    if (tid < n_w) w_mine = a_out.cnn_w[tid];
    if (summary_after_barrier) {
        // General shape: every obstacle's distance is in obst[] now (the circles' from phase 1, the polygons'
        // parked by the jobs of task 0, possibly on several waves).  The nearest obstacle of an env, lowest id
        // among equals: a wave per env when there are enough obstacles to share out, else a lane per env.
        if (O < 64) {
            // a group of lanes per env (the next power of two above its obstacles), several envs per wave
            int group = 1;
            while (group < O) group <<= 1;
            const int per_wave = 64 / group, o = lane & (group - 1);
            for (int g0 = wave * per_wave; g0 < ne; g0 += n_waves * per_wave) {
                const int el = g0 + lane / group;
                const bool have = el < ne && o < O;
                double bd = have ? obst[OB_DIST * ob_n + (have ? el * O + o : 0)] : __builtin_inf();
                int bi = bd < __builtin_inf() ? o : -1;
                for (int d = group >> 1; d >= 1; d >>= 1) {
                    const double od = __shfl_xor(bd, d, 64);
                    const int oi = __shfl_xor(bi, d, 64);
                    if (od < bd || (od == bd && oi >= 0 && (bi < 0 || oi < bi))) {
                        bd = od;
                        bi = oi;
                    }
                }
                if (el < ne && o == 0) env_summary(a_out, e0 + el, bd, bi);
            }
        } else {
            for (int el = wave; el < ne; el += n_waves) {
                double bd = __builtin_inf();
                int bi = -1;
                for (int o = lane; o < O; o += 64) {
                    const double d = obst[OB_DIST * ob_n + el * O + o];
                    if (d < bd) {
                        bd = d;
                        bi = o;
                    }
                }
                for (int d = 32; d >= 1; d >>= 1) {
                    const double od = __shfl_xor(bd, d, 64);
                    const int oi = __shfl_xor(bi, d, 64);
                    if (od < bd || (od == bd && oi >= 0 && (bi < 0 || oi < bi))) {
                        bd = od;
                        bi = oi;
                    }
                }
                if (lane == 0) env_summary(a_out, e0 + el, bd, bi);
            }
        }
        // an attached network's input takes the bytes of obst[] below: not before every wave has read them
        if (a_out.cnn_w) __syncthreads();
    }
    // sweep 1
#pragma unroll
    for (int j = 0; j < kKeep; ++j)
        if (kept[j].ok() && best[kept[j].slot()] == kept[j].key) atomicMin(&ids[kept[j].slot()], kept[j].oid());
    auto sweep1 = [&](int u, int part, bool kept_first) {
        const Unit U = unit_open(u);
        for (int j0 = (part + (kept_first ? kKeep * n_share : 0)) * 64; j0 < U.total; j0 += n_share * 64) {
            const PairHit h = unit_pair(U, j0 + lane);
            if (h.ok && best[h.slot] == h.key) atomicMin(&ids[h.slot], h.oid);
        }
    };
    if constexpr (!QUEUED) {
        if (walk_again) {
            bool first_unit = true;
            for (int w = wave - first_ray_wave; w < n_units * n_share; w += n_ray_waves) {
                const int u = FIXED ? w : w / n_share, part = FIXED ? 0 : w - u * n_share;
                sweep1(u, part, first_unit);
                first_unit = false;
            }
        }
    } else {
        // every sweep item again, off the second counter, by whichever wave is free: the rounds its first wave
        // did not keep in registers
        for (int w = wave; w < n_sweep; w = draw(1, 0)) {
            const int f = item_flag[w];
            if (!(f & 1)) continue;
            const int u = w / n_share;
            sweep1(u, w - u * n_share, (f & 2) != 0);
        }
    }
    if (by_four) {
        for (int q = tid; q < (ne * R) >> 2; q += nthr) {
            const auv_u64x2 b01 = reinterpret_cast<const auv_u64x2*>(best)[2 * q];
            const auv_u64x2 b23 = reinterpret_cast<const auv_u64x2*>(best)[2 * q + 1];
            const auv_f64x2 t01 = {__longlong_as_double((long long)b01.x), __longlong_as_double((long long)b01.y)};
            const auv_f64x2 t23 = {__longlong_as_double((long long)b23.x), __longlong_as_double((long long)b23.y)};
            const auv_f32x4 close = {(float)(1.0 - div_by_const(t01.x, range, a_out.inv_range)),
                                     (float)(1.0 - div_by_const(t01.y, range, a_out.inv_range)),
                                     (float)(1.0 - div_by_const(t23.x, range, a_out.inv_range)),
                                     (float)(1.0 - div_by_const(t23.y, range, a_out.inv_range))};
            const size_t out = (size_t)e0 * R + 4 * (size_t)q;
            __builtin_nontemporal_store(t01, reinterpret_cast<auv_f64x2*>(a_out.ranges + out));
            __builtin_nontemporal_store(t23, reinterpret_cast<auv_f64x2*>(a_out.ranges + out + 2));
            __builtin_nontemporal_store(close, reinterpret_cast<auv_f32x4*>(a_out.perception + out));
            if (a_out.cnn_w) reinterpret_cast<auv_f32x4*>(cnn_x)[q] = close;
        }
    } else {
        for (int f = tid; f < ne * R; f += nthr) {
            const double t = __longlong_as_double((long long)best[f]);
            const size_t out = (size_t)e0 * R + f;
            const float close = (float)(1.0 - div_by_const(t, range, a_out.inv_range));
            __builtin_nontemporal_store(t, &a_out.ranges[out]);
            __builtin_nontemporal_store(close, &a_out.perception[out]);
            if (a_out.cnn_w) cnn_x[f] = close;
        }
    }
    if (a_out.cnn_w) {
        if (tid < n_w) cnn_wl[tid] = w_mine;
        for (int i = tid + nthr; i < n_w; i += nthr) cnn_wl[i] = a_out.cnn_w[i];
    }
    AUV_STAMP(7);
    __syncthreads();
    AUV_STAMP(8);
    if (by_four) {
        for (int q = tid; q < (ne * R) >> 2; q += nthr)
            __builtin_nontemporal_store(reinterpret_cast<const auv_u32x4*>(ids)[q],
                                        reinterpret_cast<auv_u32x4*>(a_out.hit + (size_t)e0 * R) + q);
    } else {
        for (int f = tid; f < ne * R; f += nthr) __builtin_nontemporal_store((int)ids[f], &a_out.hit[(size_t)e0 * R + f]);
    }
    if (a_out.cnn_w) {
        // the tile's scans together, output elements dealt over the workgroup;
        // each is summed by one thread in the order k_cnn uses: same values.
        // The activation buffers take the bytes of best[], read for the last time above.
        float* act = reinterpret_cast<float*>(best);
        if (FIXED && a_out.cnn_kind == CNN_SHALLOW && a_out.cnn_head == HEAD_FEATURES) {
            // One layer of 12 outputs per scan: the last wave (it has the fewest rays to store above) forms each
            // output whole, its three sector sums in registers: a third of the instructions of the staged form
            // (no index arithmetic per sector sum, no second pass) and no further barrier.  Same operations in
            // the same order as every other evaluation of this layer.  (The compile-time shape only: in the
            // general kernels the second form costs more in scalar registers than it saves.)
            if (wave == n_waves - 1)
                conv_sectors<false, false>(cnn_x, R, act, act + cnn_p_floats(CNN_SHALLOW), cnn_act_floats(CNN_SHALLOW), ne,
                                           cnn_wl, cnn_wl + 45, 1, lane, 64, a_out.cnn_out + (size_t)e0 * 12);
        } else {
            cnn_forward<false, true>(cnn_x, R, act, act + cnn_p_floats(a_out.cnn_kind), cnn_act_floats(a_out.cnn_kind), ne,
                                     cnn_wl, a_out.cnn_out + (size_t)e0 * cnn_out_dim(a_out.cnn_kind, a_out.cnn_head),
                                     a_out.cnn_kind, a_out.cnn_head, tid, nthr);
        }
    }
    AUV_STAMP(9);
    AUV_STAMP_REAL(15);
}

// generic shape
template <int RISK>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6))) void k_sense(SenseArgs a) {
    sense_tile<RISK, false>(a);
}

// general shape, its work drawn from the queue
template <int RISK>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6))) void k_sense_queued(SenseArgs a) {
    sense_tile<RISK, false, true>(a);
}

// Which tiles the queue pays for: three rounds of 64 obstacles or more per tile, or rays enough for a unit's rounds
// to be dealt in parts.
inline bool sense_wants_queue(int TE, int O, int R) { return (TE * O + 63) / 64 >= 3 || R >= 360; }

// DefaultShape.  A CU admits floor(800 / (scalar registers rounded up to 16, + 16)) workgroups of 256
// threads, 512 / (vector registers rounded up to 8) waves per SIMD and 160 KiB of LDS: held to 80 scalar
// registers, with its 60-odd vector registers and its 19.4 KiB LDS image, the kernel gets 8 workgroups per CU,
// which is every wave slot the CU has.
template <int RISK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_sense_default(SenseArgs a) {
    sense_tile<RISK, true>(a);
}

}  // namespace auv
