// libauvsim.so: C ABI (include/auvsim.h) over the HIP kernels in
// auv_kernels.hpp.  Host side only owns buffers, validates shapes before any
// launch, and enqueues kernels on the caller's stream; there is no CPU
// fallback: without a usable gfx950 device auv_create fails.
#include "../../include/auvsim.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "auv_kernels.hpp"
#include "auv_lidar_raster.hpp"
#include "auv_sense.hpp"
#include "auv_cnn.hpp"
#include "auv_scenegen.hpp"
#include "auv_format.hpp"
#include "auv_stats.hpp"
#include "auv_train.hpp"
#include "auv_format_device.hpp"
#include "auv_dataset_rows.hpp"

using namespace auv;

namespace {

thread_local std::string g_create_error;

struct BufDesc {
    void* ptr = nullptr;
    int64_t shape[4] = {0, 0, 0, 0};
    int32_t ndim = 0;
    int32_t dtype = AUV_F64;
    size_t bytes = 0;
};

size_t dtype_size(int32_t dt) {
    switch (dt) {
        case AUV_F64: return 8;
        case AUV_F32: return 4;
        case AUV_I32: return 4;
        case AUV_U8: return 1;
        case AUV_I64: return 8;
    }
    return 0;
}

constexpr size_t kLdsMax = 160u * 1024u;  // LDS of one gfx950 CU; a workgroup may claim all of it
constexpr int kProfRing = 256;  // steps whose events are kept before a forced drain

}  // namespace

// Records of k_stats_partial for auv_episode_stats_static: a fixed array in the device image instead of an
// allocation.  In host code this name is only a handle for hipGetSymbolAddress; kernels and copies get the
// per-device address kept in g_scratch_ptr.
constexpr int kScratchRecords = AUV_STATS_MAX_ENVS / auv::kStatsEnvs;
static __device__ __align__(256) double scratch[(size_t)kScratchRecords * auv::kStatsFields];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
double* g_scratch_ptr[kMaxDevices] = {};  // address of `scratch` on each device, resolved once per device

// Resolves the address of `scratch` on `device`, which must be the current device.
hipError_t resolve_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return hipSuccess;
    double* p = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = p;
    return hipSuccess;
}
}  // namespace

struct auv_sim {
    auv_config cfg;
    Dims d;
    int device = 0;
    BufDesc buf[AUV_BUF_COUNT];
    // scene / auxiliary device memory not lent out
    double* poly_verts = nullptr;
    int32_t* poly_nv = nullptr;
    double* poly_heading = nullptr;
    double* poly_rot = nullptr;   // [V,P,NV,2] vertices turned to the ship's heading (world-frame offsets from its centre; a scene constant)
    double* poly_beam = nullptr;  // [V,P] extent of the body-frame vertices across the heading (a scene constant)
    double* poly_reach2 = nullptr;  // [V,P] squared distance of the farthest vertex from the pose point, -1 unless the
                                    // polygon is strictly convex and counter-clockwise (polygon_reach2; a scene constant)
    double* path = nullptr;        // [V,KS,kPathRec] segment records of every env's path
    int32_t* path_nseg = nullptr;  // [V] segments in use
    int KS = 1;                    // segment slots per path
    double* ray_dirs = nullptr;
    double* init_stage = nullptr;  // [6,V] staging for reset
    double* circles0 = nullptr;    // obstacle poses as uploaded (restored on reset)
    double* poly_pose0 = nullptr;
    double* envrec = nullptr;      // [V,8] k_vessel -> k_sense hand-off
    uint8_t* mask_stage = nullptr;
    double* stats_ws = nullptr;    // [stats_blocks(V), kStatsFields] records of k_stats_partial
    bool have_rays = false, have_scene = false, have_state = false;
    bool custom_rays = false;  // the ray table is not the evenly spaced full circle the rasterising path indexes by
    bool mode_forced_by_rays = false;  // lidar_mode is ALL_PAIRS only because of such a table, not by the caller's choice
    int32_t scene_upto = 0;    // envs [0, scene_upto) have been uploaded (auv_set_scene sweeps front to back)
    // launch geometry
    int lidar_te = 1, lidar_block = 256, enc_eb = 1;
    size_t lidar_lds = 0, enc_lds = 0;
    int lidar_mode = AUV_LIDAR_RASTER;
    int raster_te = 1, sense_block = 256;
    int raster_want = 0;  // envs per workgroup asked for (0 = automatic)
    int block_want = 0;   // threads per workgroup asked for (0 = automatic)
    int auto_reset = 0;   // done envs restart inside the next step
    int regenerate = 0;   // ... with a newly generated scene
    bool heading_stale = true;  // the heading_ts column has to be rewritten for every env by the next sensing pass
    int meta_every = 1;   // steps between refreshes of bearing / course / risk (0: only outside steps)
    int risk_model = AUV_RISK_REFERENCE;
    long long step_count = 0;
    bool have_generator = false, have_init = false;
    uint64_t gen_seed = 0;
    auv_scene_params gen_params;
    size_t raster_lds = 0;
    // attached LiDAR-CNN: a private copy of its weights, evaluated with every sensing pass
    float* cnn_w = nullptr;
    int cnn_kind = 0, cnn_head = 0;
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> events;  // kProfRing * 4
    int prof_pending = 0;
    double prof_ms[3] = {0, 0, 0};
    int64_t prof_steps = 0;
    std::string err;
};

namespace {

int fail(auv_sim* h, int code, const char* fmt, ...) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (h) h->err = tmp; else g_create_error = tmp;
    return code;
}

#define HIP_TRY(h, expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(h, e_ == hipErrorOutOfMemory ? AUV_ERR_MEMORY : AUV_ERR_DEVICE,       \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                       \
    } while (0)

int alloc_buf(auv_sim* h, int which, int32_t dtype, std::initializer_list<int64_t> shape) {
    BufDesc& b = h->buf[which];
    b.dtype = dtype;
    b.ndim = (int32_t)shape.size();
    size_t n = 1;
    int k = 0;
    for (int64_t s : shape) {
        b.shape[k++] = s;
        n *= (size_t)s;
    }
    b.bytes = n * dtype_size(dtype);
    // zero-sized slots (e.g. no polygons) still get a valid, distinct pointer
    HIP_TRY(h, hipMalloc(&b.ptr, b.bytes ? b.bytes : 256));
    HIP_TRY(h, hipMemset(b.ptr, 0, b.bytes ? b.bytes : 256));
    return AUV_OK;
}

template <typename T>
int alloc_raw(auv_sim* h, T** p, size_t n) {
    HIP_TRY(h, hipMalloc((void**)p, n ? n * sizeof(T) : 256));
    HIP_TRY(h, hipMemset(*p, 0, n ? n * sizeof(T) : 256));
    return AUV_OK;
}

template <typename T>
T* bp(auv_sim* h, int which) {
    return static_cast<T*>(h->buf[which].ptr);
}

// NULL is HIP's default (null) stream, exactly as in the HIP API itself
hipStream_t pick(auv_sim*, void* stream) { return (hipStream_t)stream; }

// Tile of k_sense: `want` envs per workgroup, or (want <= 0) as many as keep
// the LDS image within 24 KiB (6 workgroups per CU; 8 with the narrow image of
// the compile-time shape) while the grid still has >= 2048 workgroups, one for
// every workgroup slot of the device.
void set_raster_tile(auv_sim* h, int want) {
    const Dims& d = h->d;
    const int S = d.P * d.NV;
    const int cf = (h->cnn_w && cnn_fits_sense_kernel(h->cnn_kind)) ? h->cnn_kind : -1;
    auto bytes = [&](int te) { return sense_layout(te, d.R, d.C, d.P, d.NV, cf).total; };
    h->raster_want = want;
    int te = want;
    if (te <= 0) {
        te = 16;
        while (te > 1 && bytes(te) > 24u * 1024u) te >>= 1;
        while (te > 1 && (d.V + te - 1) / te < 2048) te >>= 1;
    }
    while (te > 1 && bytes(te) > kLdsMax) te >>= 1;
    h->raster_te = te;
    h->raster_lds = sense_layout(te, d.R, d.C, d.P, d.NV, cf).total;
    // waves per workgroup, from measurements: two for a tile of a few dozen primitives, four for the usual
    // couple of hundred, twelve for the thousand of a dense scene
    const long long prims = (long long)te * (d.C + S);
    h->sense_block = h->block_want ? h->block_want : (prims >= 1024 ? 768 : prims >= 512 ? 512 : prims > 64 ? 256 : 128);
    // the sweep carries first ray and primitive in 16 bits each, and the env of the tile in 8
    if ((long long)te * (d.C + S) > 65535 || d.R > 65535 || te > 255) h->raster_lds = (size_t)1 << 30;
}

void choose_geometry(auv_sim* h) {
    const Dims& d = h->d;
    const int S = d.P * d.NV;
    // LiDAR tile: as many envs as fit a 40 KiB LDS budget (keeps >= 3
    // workgroups per CU resident), at most 16, and few enough that small
    // batches still spread over the 256 CUs.
    const size_t per_env = lidar_lds_doubles(1, d.C, S) * sizeof(double);
    int te = (int)((40u * 1024u) / per_env);
    if (te > 16) te = 16;
    while (te > 1 && (d.V + te - 1) / te < 1024) te >>= 1;
    if (te < 1) te = 1;
    h->lidar_te = te;
    h->lidar_lds = lidar_lds_doubles(te, d.C, S) * sizeof(double);
    // whole waves of work per pass when the tile allows it
    const long long rays = (long long)te * d.R;
    h->lidar_block = (rays % 192 == 0 && rays % 256 != 0) ? 192 : 256;
    if (rays < 256) h->lidar_block = (int)(((rays + 63) / 64) * 64);
    // rasterising LiDAR: tile by its own (larger) LDS image, 36 KiB budget
    set_raster_tile(h, 0);
    // encounter: one thread per (env, obstacle), whole envs per block
    const int O = d.O > 0 ? d.O : 1;
    int eb = 256 / O;
    if (eb < 1) eb = 1;
    h->enc_eb = eb;
    h->enc_lds = (size_t)eb * O * sizeof(double);
}

// scenario generator: every env (only_done == nullptr) or the envs about to restart
int launch_scenegen(auv_sim* h, hipStream_t st, const uint8_t* only_done) {
    const Dims& d = h->d;
    SceneGenArgs ga;
    ga.d = d;
    ga.seed = h->gen_seed;
    ga.only_done = only_done;
    ga.episode = bp<int32_t>(h, AUV_BUF_EPISODE);
    ga.circles0 = h->circles0;
    ga.poly_pose0 = h->poly_pose0;
    ga.poly_verts = h->poly_verts;
    ga.poly_nv = h->poly_nv;
    ga.poly_heading = h->poly_heading;
    ga.poly_beam = h->poly_beam;
    ga.poly_reach2 = h->poly_reach2;
    ga.poly_rot = h->poly_rot;
    ga.path = h->path;
    ga.path_nseg = h->path_nseg;
    ga.KS = h->KS;
    ga.init = h->init_stage;
    const auv_scene_params& p = h->gen_params;
    ga.extent = p.extent;
    ga.r_lo = p.r_lo;
    ga.r_hi = p.r_hi;
    ga.w_lo = p.w_lo;
    ga.w_hi = p.w_hi;
    ga.s_lo = p.s_lo;
    ga.s_hi = p.s_hi;
    ga.clearance = p.clearance;
    ga.path_length = p.path_length;
    ga.size_law = p.size_law;
    const long long work = (long long)d.V * (1 + d.C + d.P);
    hipLaunchKernelGGL(k_scenegen, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, ga);
    HIP_TRY(h, hipGetLastError());
    return AUV_OK;
}

// the attached network as a launch of its own, on the perception buffer
int launch_cnn(auv_sim* h, hipStream_t st) {
    CnnArgs ca;
    ca.scans = bp<float>(h, AUV_BUF_PERCEPTION);
    ca.w = h->cnn_w;
    ca.out = bp<float>(h, AUV_BUF_CNN_OUT);
    ca.B = h->d.V;
    ca.kind = h->cnn_kind;
    ca.head = h->cnn_head;
    if (ca.kind == CNN_DEEP)
        hipLaunchKernelGGL(k_cnn_deep, dim3((unsigned)((h->d.V + kCnnScans - 1) / kCnnScans)), dim3(256),
                           cnn_deep_lds_bytes(), st, ca);
    else
        hipLaunchKernelGGL(k_cnn, dim3((unsigned)((h->d.V + kCnnScans - 1) / kCnnScans)), dim3(256),
                           cnn_lds_bytes(ca.kind), st, ca);
    HIP_TRY(h, hipGetLastError());
    return AUV_OK;
}

// Fused path: k_vessel (integration when do_step, then the vessel-only
// outputs) followed by k_sense (scene pass).  `action` is only read when
// do_step is set.
int launch_fused(auv_sim* h, hipStream_t st, hipEvent_t* ev, int do_step, const double* action) {
    const Dims& d = h->d;
    if (do_step && h->auto_reset && h->regenerate) {
        // new scenes for the envs that k_vessel is about to restart
        int rc = launch_scenegen(h, st, bp<uint8_t>(h, AUV_BUF_DONE));
        if (rc) return rc;
    }
    VesselArgs va;
    va.V = d.V;
    va.do_step = do_step;
    va.auto_reset = h->auto_reset;
    va.done = bp<uint8_t>(h, AUV_BUF_DONE);
    va.init = h->init_stage;
    va.episode = bp<int32_t>(h, AUV_BUF_EPISODE);
    va.C = d.C;
    va.P = d.P;
    va.circles = bp<double>(h, AUV_BUF_CIRCLES);
    va.circles0 = h->circles0;
    va.poly_pose = bp<double>(h, AUV_BUF_POLY_POSE);
    va.poly_pose0 = h->poly_pose0;
    va.max_steps = h->cfg.max_steps;
    va.state = bp<double>(h, AUV_BUF_STATE);
    va.action = action;
    va.t_step = bp<int32_t>(h, AUV_BUF_T_STEP);
    va.path = h->path;
    va.path_nseg = h->path_nseg;
    va.path_seg = bp<int32_t>(h, AUV_BUF_PATH_SEG);
    va.KS = h->KS;
    va.envrec = h->envrec;
    va.own = bp<double>(h, AUV_BUF_OWN);
    va.along = bp<double>(h, AUV_BUF_ALONG);
    va.navigation = bp<float>(h, AUV_BUF_NAVIGATION);
    va.reached_goal = bp<uint8_t>(h, AUV_BUF_REACHED_GOAL);
    va.truncated = bp<uint8_t>(h, AUV_BUF_TRUNCATED);
    va.dt = h->cfg.dt;
    va.thrust_max = h->cfg.thrust_max;
    va.moment_max = h->cfg.moment_max;
    va.look_ahead = h->cfg.look_ahead;
    va.goal_radius = h->cfg.goal_radius;
    va.rw_progress = h->cfg.reward[0];
    va.rw_cte = h->cfg.reward[3];
    for (int i = 0; i < MP_COUNT; ++i) va.mp[i] = h->cfg.model[i];
    hipLaunchKernelGGL(k_vessel, dim3((d.V + 255) / 256), dim3(256), 0, st, va);
    HIP_TRY(h, hipGetLastError());
    if (ev) {
        HIP_TRY(h, hipEventRecord(ev[1], st));
        HIP_TRY(h, hipEventRecord(ev[2], st));  // no separate encounter kernel on this path
    }
    SenseArgs sa;
    sa.d = d;
    sa.TE = h->raster_te;
    sa.is_step = do_step;
    if (do_step) h->step_count++;
    sa.with_meta = !do_step || (h->meta_every > 0 && h->step_count % h->meta_every == 0);
    // an obstacle's heading_ts is a constant of the scene: rewritten outside steps, after the scene or the risk
    // index changed, and (in the kernel) for envs that restart in this step
    sa.refresh_heading = !do_step || h->heading_stale || h->meta_every != 1;
    if (sa.with_meta) h->heading_stale = false;
    sa.state = bp<double>(h, AUV_BUF_STATE);
    sa.envrec = h->envrec;
    sa.circles = bp<double>(h, AUV_BUF_CIRCLES);
    sa.poly_pose = bp<double>(h, AUV_BUF_POLY_POSE);
    sa.poly_verts = h->poly_verts;
    sa.poly_nv = h->poly_nv;
    sa.poly_heading = h->poly_heading;
    sa.poly_beam = h->poly_beam;
    sa.poly_reach2 = h->poly_reach2;
    sa.poly_rot = h->poly_rot;
    sa.ray_dirs = h->ray_dirs;
    sa.ranges = bp<double>(h, AUV_BUF_RANGES);
    sa.hit = bp<int32_t>(h, AUV_BUF_HIT);
    sa.perception = bp<float>(h, AUV_BUF_PERCEPTION);
    sa.obst_dist = bp<double>(h, AUV_BUF_OBST_DIST);
    sa.risk = bp<double>(h, AUV_BUF_RISK);
    sa.meta = bp<double>(h, AUV_BUF_META);
    sa.collision = bp<uint8_t>(h, AUV_BUF_COLLISION);
    sa.reached_goal = bp<uint8_t>(h, AUV_BUF_REACHED_GOAL);
    sa.truncated = bp<uint8_t>(h, AUV_BUF_TRUNCATED);
    sa.done = bp<uint8_t>(h, AUV_BUF_DONE);
    sa.min_dist = bp<double>(h, AUV_BUF_MIN_DIST);
    sa.min_idx = bp<int32_t>(h, AUV_BUF_MIN_IDX);
    sa.reward = bp<double>(h, AUV_BUF_REWARD);
    sa.dt = h->cfg.dt;
    sa.sensor_range = h->cfg.sensor_range;
    sa.inv_range = 1.0 / h->cfg.sensor_range;
    sa.vessel_radius = h->cfg.vessel_radius;
    sa.contact_dist = h->cfg.contact_dist;
    sa.w_d = h->cfg.risk[0];
    sa.w_c = h->cfg.risk[1];
    sa.t_c = h->cfg.risk[2];
    sa.rw_collision = h->cfg.reward[1];
    sa.rw_goal = h->cfg.reward[2];
    const bool cnn_inside = h->cnn_w && cnn_fits_sense_kernel(h->cnn_kind);
    sa.cnn_w = cnn_inside ? h->cnn_w : nullptr;
    sa.cnn_out = bp<float>(h, AUV_BUF_CNN_OUT);
    sa.cnn_kind = h->cnn_kind;
    {
        const SenseLayout L = sense_layout(sa.TE, d.R, d.C, d.P, d.NV, sa.cnn_w ? sa.cnn_kind : -1);
        const size_t off[10] = {L.env, L.circ, L.seg, L.best, L.obst, L.edist, L.ids, L.ecross, L.live, L.wsum};
        for (int i = 0; i < 10; ++i) sa.lds[i] = (int)off[i];
    }
    sa.cnn_head = h->cnn_head;
    const int blocks = (d.V + h->raster_te - 1) / h->raster_te;
    const bool fixed = is_default_shape(d, h->raster_te, h->sense_block);
    // the compile-time shape has a narrower LDS image
    const size_t lds_fixed = sense_layout(sa.TE, d.R, d.C, d.P, d.NV, sa.cnn_w ? sa.cnn_kind : -1, true).total;
    const bool queued = !fixed && sense_wants_queue(sa.TE, d.O, d.R);
    if (h->risk_model == AUV_RISK_REFERENCE) {
        if (fixed) hipLaunchKernelGGL(k_sense_default<1>, dim3(blocks), dim3(h->sense_block), lds_fixed, st, sa);
        else if (queued) hipLaunchKernelGGL(k_sense_queued<1>, dim3(blocks), dim3(h->sense_block), h->raster_lds, st, sa);
        else hipLaunchKernelGGL(k_sense<1>, dim3(blocks), dim3(h->sense_block), h->raster_lds, st, sa);
    } else {
        if (fixed) hipLaunchKernelGGL(k_sense_default<0>, dim3(blocks), dim3(h->sense_block), lds_fixed, st, sa);
        else if (queued) hipLaunchKernelGGL(k_sense_queued<0>, dim3(blocks), dim3(h->sense_block), h->raster_lds, st, sa);
        else hipLaunchKernelGGL(k_sense<0>, dim3(blocks), dim3(h->sense_block), h->raster_lds, st, sa);
    }
    HIP_TRY(h, hipGetLastError());
    if (h->cnn_w && !cnn_inside) {
        int rc = launch_cnn(h, st);  // the large network keeps a launch of its own
        if (rc) return rc;
    }
    if (ev) HIP_TRY(h, hipEventRecord(ev[3], st));
    return AUV_OK;
}

// Sensing on the all-pairs (cross-check) path, or, for a call that is not part
// of a step, on either path.
int launch_sense(auv_sim* h, hipStream_t st, hipEvent_t* ev, int is_step) {
    if (h->lidar_mode == AUV_LIDAR_RASTER) return launch_fused(h, st, ev, 0, nullptr);
    const Dims& d = h->d;
    EncArgs ea;
    ea.d = d;
    ea.EB = h->enc_eb;
    ea.state = bp<double>(h, AUV_BUF_STATE);
    ea.t_step = bp<int32_t>(h, AUV_BUF_T_STEP);
    ea.circles = bp<double>(h, AUV_BUF_CIRCLES);
    ea.poly_pose = bp<double>(h, AUV_BUF_POLY_POSE);
    ea.poly_verts = h->poly_verts;
    ea.poly_nv = h->poly_nv;
    ea.poly_heading = h->poly_heading;
    ea.path = h->path;
    ea.path_nseg = h->path_nseg;
    ea.path_seg = bp<int32_t>(h, AUV_BUF_PATH_SEG);
    ea.KS = h->KS;
    ea.obst_dist = bp<double>(h, AUV_BUF_OBST_DIST);
    ea.risk = bp<double>(h, AUV_BUF_RISK);
    ea.meta = bp<double>(h, AUV_BUF_META);
    ea.own = bp<double>(h, AUV_BUF_OWN);
    ea.collision = bp<uint8_t>(h, AUV_BUF_COLLISION);
    ea.reached_goal = bp<uint8_t>(h, AUV_BUF_REACHED_GOAL);
    ea.truncated = bp<uint8_t>(h, AUV_BUF_TRUNCATED);
    ea.done = bp<uint8_t>(h, AUV_BUF_DONE);
    ea.min_dist = bp<double>(h, AUV_BUF_MIN_DIST);
    ea.min_idx = bp<int32_t>(h, AUV_BUF_MIN_IDX);
    ea.along = bp<double>(h, AUV_BUF_ALONG);
    ea.reward = bp<double>(h, AUV_BUF_REWARD);
    ea.navigation = bp<float>(h, AUV_BUF_NAVIGATION);
    ea.sensor_range = h->cfg.sensor_range;
    ea.vessel_radius = h->cfg.vessel_radius;
    ea.contact_dist = h->cfg.contact_dist;
    ea.look_ahead = h->cfg.look_ahead;
    ea.goal_radius = h->cfg.goal_radius;
    ea.w_d = h->cfg.risk[0];
    ea.w_c = h->cfg.risk[1];
    ea.t_c = h->cfg.risk[2];
    ea.max_steps = h->cfg.max_steps;
    ea.rw_progress = h->cfg.reward[0];
    ea.rw_collision = h->cfg.reward[1];
    ea.rw_goal = h->cfg.reward[2];
    ea.rw_cte = h->cfg.reward[3];
    ea.is_step = is_step;
    ea.risk_model = h->risk_model;
    const int enc_blocks = (d.V + h->enc_eb - 1) / h->enc_eb;
    hipLaunchKernelGGL(k_encounter, dim3(enc_blocks), dim3(256), h->enc_lds, st, ea);
    HIP_TRY(h, hipGetLastError());
    if (ev) HIP_TRY(h, hipEventRecord(ev[2], st));

    LidarArgs la;
    la.d = d;
    la.TE = h->lidar_te;
    la.state = bp<double>(h, AUV_BUF_STATE);
    la.circles = bp<double>(h, AUV_BUF_CIRCLES);
    la.poly_pose = bp<double>(h, AUV_BUF_POLY_POSE);
    la.poly_verts = h->poly_verts;
    la.poly_nv = h->poly_nv;
    la.poly_reach2 = h->poly_reach2;
    la.ray_dirs = h->ray_dirs;
    la.ranges = bp<double>(h, AUV_BUF_RANGES);
    la.hit = bp<int32_t>(h, AUV_BUF_HIT);
    la.perception = bp<float>(h, AUV_BUF_PERCEPTION);
    la.sensor_range = h->cfg.sensor_range;
    const int lid_blocks = (d.V + h->lidar_te - 1) / h->lidar_te;
    hipLaunchKernelGGL(k_lidar, dim3(lid_blocks), dim3(h->lidar_block), h->lidar_lds, st, la);
    HIP_TRY(h, hipGetLastError());
    if (h->cnn_w) {
        int rc = launch_cnn(h, st);
        if (rc) return rc;
    }
    if (ev) HIP_TRY(h, hipEventRecord(ev[3], st));
    return AUV_OK;
}

// the two launches of the statistics reduction, with `records` as the workspace
int launch_stats(auv_sim* h, double* records, double* out, int accumulate, hipStream_t st) {
    const Dims& d = h->d;
    StatsArgs sa;
    sa.V = d.V;
    sa.R = d.R;
    sa.O = d.O;
    sa.done = bp<uint8_t>(h, AUV_BUF_DONE);
    sa.collision = bp<uint8_t>(h, AUV_BUF_COLLISION);
    sa.reached_goal = bp<uint8_t>(h, AUV_BUF_REACHED_GOAL);
    sa.truncated = bp<uint8_t>(h, AUV_BUF_TRUNCATED);
    sa.reward = bp<double>(h, AUV_BUF_REWARD);
    sa.ranges = bp<double>(h, AUV_BUF_RANGES);
    sa.risk = bp<double>(h, AUV_BUF_RISK);
    sa.records = records;
    sa.envs_per_block = stats_envs_per_block(d.V);
    const int blocks = stats_blocks(d.V);
    if (blocks > kScratchRecords && d.V <= AUV_STATS_MAX_ENVS)  // the fixed array holds every batch it admits
        return fail(h, AUV_ERR_STATE, "statistics grid of %d workgroups exceeds %d records", blocks, kScratchRecords);
    hipLaunchKernelGGL(k_stats_partial, dim3(blocks), dim3(kStatsThreads), 0, st, sa);
    HIP_TRY(h, hipGetLastError());
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(kStatsThreads), 0, st, records, blocks, out, accumulate);
    HIP_TRY(h, hipGetLastError());
    return AUV_OK;
}

// Segment records of envs [env_begin, env_begin + env_count): `segments` is host f64 [n, ks, 6] = x0, y0, angle,
// cos, sin, length per segment (ks <= KS slots), `n_segments` host i32 [n] or null (= ks everywhere).  The
// arc length before each segment is summed here, in segment order.
int upload_path(auv_sim* h, int32_t env_begin, int32_t env_count, const double* segments, int ks, const int32_t* n_segments) {
    const size_t n = (size_t)env_count, KS = (size_t)h->KS;
    std::vector<double> tab(n * KS * kPathRec, 0.0);
    std::vector<int32_t> cnt(n, ks);
    for (size_t i = 0; i < n; ++i) {
        if (n_segments) {
            if (n_segments[i] < 1 || n_segments[i] > ks)
                return fail(h, AUV_ERR_ARG, "path of env %zu has %d segments, expected 1..%d", (size_t)env_begin + i, n_segments[i], ks);
            cnt[i] = n_segments[i];
        }
        double cum = 0.0;
        for (int k = 0; k < ks; ++k) {
            const double* src = segments + (i * (size_t)ks + k) * 6;
            double* dst = &tab[(i * KS + k) * kPathRec];
            for (int f = 0; f < 6; ++f) dst[f] = src[f];
            if (k < cnt[i]) {
                // A vessel leaves a segment when its projection exceeds the segment's length: on a segment
                // without length, or whose direction is not a unit vector, it would stay for ever.
                const double uu = src[3] * src[3] + src[4] * src[4];
                if (!(src[5] > 0.0) || !(std::fabs(uu - 1.0) <= 1e-9))
                    return fail(h, AUV_ERR_ARG, "path of env %zu, segment %d: length %g and direction (%g, %g); a segment "
                                "needs a length > 0 and (cos, sin) of its angle (drop repeated waypoints)",
                                (size_t)env_begin + i, k, src[5], src[3], src[4]);
            }
            dst[6] = cum;
            cum = cum + src[5];
        }
    }
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(h->path + (size_t)env_begin * KS * kPathRec, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->path_nseg + env_begin, cnt.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    return AUV_OK;
}

int drain_profile(auv_sim* h) {
    if (h->prof_pending == 0) return AUV_OK;
    hipEvent_t* last = &h->events[(size_t)(h->prof_pending - 1) * 4];
    HIP_TRY(h, hipEventSynchronize(last[3]));
    for (int s = 0; s < h->prof_pending; ++s) {
        hipEvent_t* ev = &h->events[(size_t)s * 4];
        for (int k = 0; k < 3; ++k) {
            float ms = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            h->prof_ms[k] += ms;
        }
    }
    h->prof_steps += h->prof_pending;
    h->prof_pending = 0;
    return AUV_OK;
}

int ready_to_run(auv_sim* h, bool need_state) {
    if (!h->have_rays) return fail(h, AUV_ERR_STATE, "auv_set_rays has not been called");
    if (!h->have_scene)
        return fail(h, AUV_ERR_STATE, "auv_set_scene has not covered every env yet: envs [%d, %d) are missing "
                    "(upload front to back)", h->scene_upto, h->d.V);
    if (need_state && !h->have_state) return fail(h, AUV_ERR_STATE, "auv_reset has not been called");
    return AUV_OK;
}

int do_step(auv_sim* h, const double* action_dev, hipStream_t st) {
    const Dims& d = h->d;
    hipEvent_t* ev = nullptr;
    if (h->profiling) {
        if (h->prof_pending == kProfRing) {
            int rc = drain_profile(h);
            if (rc) return rc;
        }
        ev = &h->events[(size_t)h->prof_pending * 4];
        HIP_TRY(h, hipEventRecord(ev[0], st));
    }
    if (h->lidar_mode == AUV_LIDAR_RASTER) {
        int rc = launch_fused(h, st, ev, 1, action_dev ? action_dev : bp<double>(h, AUV_BUF_ACTION));
        if (rc) return rc;
        if (ev) h->prof_pending++;
        return AUV_OK;
    }
    DynArgs da;
    da.d = d;
    da.state = bp<double>(h, AUV_BUF_STATE);
    da.action = action_dev ? action_dev : bp<double>(h, AUV_BUF_ACTION);
    da.t_step = bp<int32_t>(h, AUV_BUF_T_STEP);
    da.circles = bp<double>(h, AUV_BUF_CIRCLES);
    da.poly_pose = bp<double>(h, AUV_BUF_POLY_POSE);
    da.dt = h->cfg.dt;
    da.thrust_max = h->cfg.thrust_max;
    da.moment_max = h->cfg.moment_max;
    for (int i = 0; i < MP_COUNT; ++i) da.mp[i] = h->cfg.model[i];
    da.advance = 1;
    const long long work = (long long)d.V * (d.O > 1 ? d.O : 1);
    long long blocks = (work + 255) / 256;
    const long long min_blocks = ((long long)d.V + 255) / 256;
    if (blocks > 65536) blocks = 65536;  // obstacle advance is grid-strided
    if (blocks < min_blocks) blocks = min_blocks;
    hipLaunchKernelGGL(k_dynamics, dim3((unsigned)blocks), dim3(256), 0, st, da);
    HIP_TRY(h, hipGetLastError());
    if (ev) HIP_TRY(h, hipEventRecord(ev[1], st));
    int rc = launch_sense(h, st, ev, 1);
    if (rc) return rc;
    if (ev) h->prof_pending++;
    return AUV_OK;
}

}  // namespace

extern "C" {

#define AUV_STR2(x) #x
#define AUV_STR(x) AUV_STR2(x)
const char* auv_version(void) { return "libauvsim abi " AUV_STR(AUVSIM_ABI_VERSION) " gfx950 fp64-unfused"; }

const char* auv_last_error(auv_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int auv_create(const auv_config* cfg, int device, auv_handle* out) {
    if (!cfg || !out) return fail(nullptr, AUV_ERR_ARG, "auv_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != AUVSIM_ABI_VERSION)
        return fail(nullptr, AUV_ERR_ARG, "auv_create: abi_version %d, library is %d", cfg->abi_version,
                    AUVSIM_ABI_VERSION);
    if (cfg->n_envs < 1 || cfg->n_rays < 1 || cfg->n_circles < 0 || cfg->n_polys < 0)
        return fail(nullptr, AUV_ERR_ARG, "auv_create: n_envs/n_rays must be >= 1, slot counts >= 0");
    if (cfg->n_polys > 0 && (cfg->n_verts < 3 || cfg->n_verts > 16))
        return fail(nullptr, AUV_ERR_ARG, "auv_create: n_verts must be in 3..16");
    if ((long long)cfg->n_envs * cfg->n_rays > (1ll << 31) - 1 ||
        (long long)cfg->n_envs * (cfg->n_circles + cfg->n_polys) > (1ll << 31) - 1)
        return fail(nullptr, AUV_ERR_ARG, "auv_create: batch too large for 32-bit ray/obstacle indices");
    if (cfg->n_path_segments < 0 || cfg->n_path_segments > 4096)
        return fail(nullptr, AUV_ERR_ARG, "auv_create: n_path_segments must be in 0..4096");
    if (!(cfg->dt > 0.0) || !(cfg->sensor_range > 0.0))
        return fail(nullptr, AUV_ERR_ARG, "auv_create: dt and sensor_range must be positive");
    if (cfg->contact_dist != cfg->contact_dist)
        return fail(nullptr, AUV_ERR_ARG, "auv_create: contact_dist is NaN");
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, AUV_ERR_DEVICE, "auv_create: no HIP device (%s)",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n_dev)
        return fail(nullptr, AUV_ERR_ARG, "auv_create: device %d out of range (have %d)", device, n_dev);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, AUV_ERR_DEVICE, "hipSetDevice: %s", hipGetErrorString(e));

    e = resolve_scratch(device);  // once per device, here, so that auv_episode_stats_static never sets anything up
    if (e != hipSuccess)
        return fail(nullptr, AUV_ERR_DEVICE, "auv_create: cannot resolve the statistics scratch array on device %d: %s",
                    device, hipGetErrorString(e));
    auv_sim* h = new auv_sim();
    h->cfg = *cfg;
    h->device = device;
    Dims& d = h->d;
    d.V = cfg->n_envs;
    d.R = cfg->n_rays;
    d.C = cfg->n_circles;
    d.P = cfg->n_polys;
    d.NV = cfg->n_polys > 0 ? cfg->n_verts : 1;
    d.O = d.C + d.P;
    h->KS = cfg->n_path_segments > 0 ? cfg->n_path_segments : 1;
    choose_geometry(h);
    // dynamic LDS beyond the 64 KiB default window has to be granted per kernel
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_sense<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsMax) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_sense<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsMax) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_sense_queued<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsMax) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_sense_queued<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsMax) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_lidar), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLdsMax) != hipSuccess) {
        delete h;
        return fail(nullptr, AUV_ERR_DEVICE, "auv_create: cannot raise the dynamic LDS limit to %zu", kLdsMax);
    }
    if (h->raster_lds > kLdsMax) h->lidar_mode = AUV_LIDAR_ALL_PAIRS;  // image too big: fall back to the smaller one
    if (h->lidar_lds > kLdsMax) {
        // one env's primitives must fit the default dynamic-LDS window
        std::string msg = "auv_create: one env needs " + std::to_string(h->lidar_lds) +
                          " B of LDS for its obstacle primitives (limit 163840)";
        delete h;
        return fail(nullptr, AUV_ERR_ARG, "%s", msg.c_str());
    }
    int rc = AUV_OK;
    auto chk = [&](int r) { if (rc == AUV_OK && r != AUV_OK) rc = r; };
    const int64_t V = d.V, R = d.R, C = d.C, P = d.P, NV = d.NV, O = d.O;
    chk(alloc_buf(h, AUV_BUF_STATE, AUV_F64, {6, V}));
    chk(alloc_buf(h, AUV_BUF_ACTION, AUV_F64, {2, V}));
    chk(alloc_buf(h, AUV_BUF_RANGES, AUV_F64, {V, R}));
    chk(alloc_buf(h, AUV_BUF_HIT, AUV_I32, {V, R}));
    chk(alloc_buf(h, AUV_BUF_PERCEPTION, AUV_F32, {V, 1, R}));
    chk(alloc_buf(h, AUV_BUF_NAVIGATION, AUV_F32, {V, 1, AUV_NAV_FEATURES}));
    chk(alloc_buf(h, AUV_BUF_OBST_DIST, AUV_F64, {V, O}));
    chk(alloc_buf(h, AUV_BUF_RISK, AUV_F64, {V, O}));
    chk(alloc_buf(h, AUV_BUF_META, AUV_F64, {V, O, AUV_META_FIELDS}));
    chk(alloc_buf(h, AUV_BUF_OWN, AUV_F64, {V, 2}));
    chk(alloc_buf(h, AUV_BUF_COLLISION, AUV_U8, {V}));
    chk(alloc_buf(h, AUV_BUF_REACHED_GOAL, AUV_U8, {V}));
    chk(alloc_buf(h, AUV_BUF_TRUNCATED, AUV_U8, {V}));
    chk(alloc_buf(h, AUV_BUF_DONE, AUV_U8, {V}));
    chk(alloc_buf(h, AUV_BUF_MIN_DIST, AUV_F64, {V}));
    chk(alloc_buf(h, AUV_BUF_MIN_IDX, AUV_I32, {V}));
    chk(alloc_buf(h, AUV_BUF_ALONG, AUV_F64, {V}));
    chk(alloc_buf(h, AUV_BUF_T_STEP, AUV_I32, {V}));
    chk(alloc_buf(h, AUV_BUF_CIRCLES, AUV_F64, {V, C, 5}));
    chk(alloc_buf(h, AUV_BUF_POLY_POSE, AUV_F64, {V, P, 6}));
    chk(alloc_buf(h, AUV_BUF_REWARD, AUV_F64, {V}));
    chk(alloc_buf(h, AUV_BUF_EPISODE, AUV_I32, {V}));
    chk(alloc_buf(h, AUV_BUF_CNN_OUT, AUV_F32, {V, 12}));  // widest head; the shape follows the attached network
    chk(alloc_raw(h, &h->poly_verts, (size_t)V * P * NV * 2));
    chk(alloc_raw(h, &h->poly_rot, (size_t)V * P * NV * 2));
    chk(alloc_raw(h, &h->poly_nv, (size_t)V * P));
    chk(alloc_raw(h, &h->poly_heading, (size_t)V * P));
    chk(alloc_raw(h, &h->poly_beam, (size_t)V * P));
    chk(alloc_raw(h, &h->poly_reach2, (size_t)V * P));
    chk(alloc_raw(h, &h->path, (size_t)V * h->KS * kPathRec));
    chk(alloc_raw(h, &h->path_nseg, (size_t)V));
    chk(alloc_buf(h, AUV_BUF_PATH_SEG, AUV_I32, {V}));
    chk(alloc_raw(h, &h->ray_dirs, (size_t)R * 2));
    chk(alloc_raw(h, &h->init_stage, (size_t)V * 6));
    chk(alloc_raw(h, &h->circles0, (size_t)V * C * 5));
    chk(alloc_raw(h, &h->poly_pose0, (size_t)V * P * 6));
    chk(alloc_raw(h, &h->envrec, (size_t)V * ER_COUNT));
    chk(alloc_raw(h, &h->mask_stage, (size_t)V));
    chk(alloc_raw(h, &h->stats_ws, (size_t)stats_blocks(d.V) * kStatsFields));
    if (rc == AUV_OK && hipDeviceSynchronize() != hipSuccess) rc = AUV_ERR_DEVICE;
    if (rc != AUV_OK) {
        g_create_error = "auv_create: " + h->err;
        auv_destroy(h);
        return rc;
    }
    // default path: along +x from the origin (cos = 1), so navigation is defined
    // even when the caller never uploads paths
    {
        std::vector<double> p((size_t)V * h->KS * kPathRec, 0.0);
        std::vector<int32_t> one((size_t)V, 1);
        for (int64_t i = 0; i < V; ++i) p[(size_t)i * h->KS * kPathRec + 3] = 1.0;
        if (hipMemcpy(h->path, p.data(), p.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->path_nseg, one.data(), one.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
            auv_destroy(h);
            return fail(nullptr, AUV_ERR_DEVICE, "auv_create: path upload failed");
        }
    }
    *out = h;
    return AUV_OK;
}

void auv_destroy(auv_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (auto& ev : h->events) (void)hipEventDestroy(ev);
    for (auto& b : h->buf)
        if (b.ptr) (void)hipFree(b.ptr);
    void* raw[] = {h->poly_verts, h->poly_rot, h->poly_nv, h->poly_heading, h->poly_beam, h->poly_reach2, h->path, h->path_nseg, h->ray_dirs, h->init_stage, h->mask_stage,
                   h->circles0, h->poly_pose0, h->envrec, h->cnn_w, h->stats_ws};
    for (void* p : raw)
        if (p) (void)hipFree(p);
    delete h;
}

int auv_set_rays(auv_handle h, const double* ray_dirs) {
    if (!h || !ray_dirs) return fail(h, AUV_ERR_ARG, "auv_set_rays: null argument");
    // The rasterising kernel picks candidate rays by index arithmetic that assumes ray i at body angle
    // -pi + (i + 1) 2 pi / R.  Any other table (a sector sensor, another start, uneven spacing) takes the
    // all-pairs path, which reads the directions only.
    const int R = h->d.R;
    bool custom = false;
    for (int i = 0; i < R && !custom; ++i) {
        const double ang = -3.141592653589793 + (i + 1.0) * (6.283185307179586 / R);
        custom = std::fabs(ray_dirs[2 * i] - std::cos(ang)) > 1e-9 || std::fabs(ray_dirs[2 * i + 1] - std::sin(ang)) > 1e-9;
    }
    if (custom && h->auto_reset)
        return fail(h, AUV_ERR_ARG, "auv_set_rays: this ray table is not the evenly spaced full circle; it needs the "
                    "all-pairs path, which has no in-step restart (clear AUV_OPT_AUTO_RESET first)");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(h->ray_dirs, ray_dirs, (size_t)R * 2 * sizeof(double), hipMemcpyHostToDevice));
    h->have_rays = true;
    h->custom_rays = custom;
    if (custom && h->lidar_mode == AUV_LIDAR_RASTER) {
        h->lidar_mode = AUV_LIDAR_ALL_PAIRS;
        h->mode_forced_by_rays = true;
    } else if (!custom && h->mode_forced_by_rays) {
        // the standard table is back: so is the path the caller had before the custom one forced the other
        if (h->raster_lds <= kLdsMax) h->lidar_mode = AUV_LIDAR_RASTER;
        h->mode_forced_by_rays = false;
    }
    return AUV_OK;
}

int auv_set_scene(auv_handle h, int32_t env_begin, int32_t env_count, const double* circles,
                  const double* poly_pose, const double* poly_verts, const int32_t* poly_nv,
                  const double* poly_heading, const double* path) {
    if (!h) return AUV_ERR_ARG;
    const Dims& d = h->d;
    if (env_begin < 0 || env_count < 1 || (long long)env_begin + env_count > d.V)
        return fail(h, AUV_ERR_ARG, "auv_set_scene: env range [%d,%d) outside [0,%d)", env_begin,
                    env_begin + env_count, d.V);
    if (d.C > 0 && !circles) return fail(h, AUV_ERR_ARG, "auv_set_scene: circles is NULL but n_circles > 0");
    if (d.P > 0 && (!poly_pose || !poly_verts || !poly_nv || !poly_heading))
        return fail(h, AUV_ERR_ARG, "auv_set_scene: a polygon array is NULL but n_polys > 0");
    if (d.P > 0) {
        const size_t n = (size_t)env_count * d.P;
        for (size_t i = 0; i < n; ++i)
            if (poly_nv[i] != 0 && (poly_nv[i] < 3 || poly_nv[i] > d.NV))
                return fail(h, AUV_ERR_ARG, "auv_set_scene: poly_nv[%zu] = %d not 0 or in 3..%d", i, poly_nv[i], d.NV);
    }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t b = (size_t)env_begin, n = (size_t)env_count;
    if (d.C > 0) {
        HIP_TRY(h, hipMemcpy(bp<double>(h, AUV_BUF_CIRCLES) + b * d.C * 5, circles, n * d.C * 5 * sizeof(double),
                             hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->circles0 + b * d.C * 5, circles, n * d.C * 5 * sizeof(double),
                             hipMemcpyHostToDevice));
    }
    if (d.P > 0) {
        HIP_TRY(h, hipMemcpy(bp<double>(h, AUV_BUF_POLY_POSE) + b * d.P * 6, poly_pose,
                             n * d.P * 6 * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->poly_pose0 + b * d.P * 6, poly_pose, n * d.P * 6 * sizeof(double),
                             hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->poly_verts + b * d.P * d.NV * 2, poly_verts,
                             n * d.P * d.NV * 2 * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->poly_nv + b * d.P, poly_nv, n * d.P * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->poly_heading + b * d.P, poly_heading, n * d.P * sizeof(double),
                             hipMemcpyHostToDevice));
        // beam of every ship: max - min of the body-frame y of its vertices
        std::vector<double> beam(n * d.P, 0.0);
        for (size_t i = 0; i < n * d.P; ++i) {
            double hi = -INFINITY, lo = INFINITY;
            for (int k = 0; k < poly_nv[i]; ++k) {
                const double y = poly_verts[(i * d.NV + k) * 2 + 1];
                hi = y > hi ? y : hi;
                lo = y < lo ? y : lo;
            }
            if (poly_nv[i] > 0) beam[i] = hi - lo;
        }
        HIP_TRY(h, hipMemcpy(h->poly_beam + b * d.P, beam.data(), n * d.P * sizeof(double), hipMemcpyHostToDevice));
        // which polygons show the sensor only their facing edges, and from how far out
        std::vector<double> reach2(n * d.P, -1.0);
        for (size_t i = 0; i < n * d.P; ++i) reach2[i] = polygon_reach2(poly_verts + i * d.NV * 2, poly_nv[i]);
        HIP_TRY(h, hipMemcpy(h->poly_reach2 + b * d.P, reach2.data(), n * d.P * sizeof(double), hipMemcpyHostToDevice));
        // every vertex slot turned to its ship's heading: R(h) * body, the offsets the sensing kernel adds to the centre
        std::vector<double> rot(n * d.P * d.NV * 2);
        for (size_t i = 0; i < n * d.P; ++i) {
            const double ch = poly_pose[i * 6 + 2], sh = poly_pose[i * 6 + 3];
            for (int k = 0; k < d.NV; ++k) {
                const double vx = poly_verts[(i * d.NV + k) * 2], vy = poly_verts[(i * d.NV + k) * 2 + 1];
                rot[(i * d.NV + k) * 2] = ch * vx - sh * vy;
                rot[(i * d.NV + k) * 2 + 1] = sh * vx + ch * vy;
            }
        }
        HIP_TRY(h, hipMemcpy(h->poly_rot + b * d.P * d.NV * 2, rot.data(), rot.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (path) {
        int rc = upload_path(h, env_begin, env_count, path, 1, nullptr);  // one straight segment per env
        if (rc) return rc;
    }
    // the scene is complete once the calls so far have covered [0, V): each call may extend the covered
    // prefix (or rewrite envs inside it)
    h->heading_stale = true;
    if (env_begin <= h->scene_upto && env_begin + env_count > h->scene_upto) h->scene_upto = env_begin + env_count;
    if (h->scene_upto == d.V) h->have_scene = true;
    return AUV_OK;
}

int auv_set_path(auv_handle h, int32_t env_begin, int32_t env_count, const double* segments, int32_t n_slots,
                 const int32_t* n_segments) {
    if (!h) return AUV_ERR_ARG;
    if (!segments) return fail(h, AUV_ERR_ARG, "auv_set_path: segments is NULL");
    if (env_begin < 0 || env_count < 1 || (long long)env_begin + env_count > h->d.V)
        return fail(h, AUV_ERR_ARG, "auv_set_path: env range [%d,%d) outside [0,%d)", env_begin, env_begin + env_count, h->d.V);
    if (n_slots < 1 || n_slots > h->KS)
        return fail(h, AUV_ERR_ARG, "auv_set_path: %d segments per env, the batch was created with room for %d "
                    "(auv_config.n_path_segments)", n_slots, h->KS);
    return upload_path(h, env_begin, env_count, segments, n_slots, n_segments);
}

int auv_generate_scene(auv_handle h, uint64_t seed, const auv_scene_params* params, void* stream) {
    if (!h) return AUV_ERR_ARG;
    const Dims& d = h->d;
    if (d.P > 0 && d.NV < 5) return fail(h, AUV_ERR_ARG, "auv_generate_scene: ships need n_verts >= 5, batch has %d", d.NV);
    auv_scene_params p = {200.0, 5.0, 30.0, 5.0, 20.0, 0.3, 1.2, 25.0, 800.0, AUV_SIZES_UNIFORM, 0};
    if (params) p = *params;
    if (!(p.extent > 0.0) || !(p.r_lo > 0.0) || !(p.r_hi >= p.r_lo) || !(p.w_lo > 0.0) || !(p.w_hi >= p.w_lo) ||
        !(p.s_lo >= 0.0) || !(p.s_hi >= p.s_lo) || !(p.clearance >= 0.0) || !(p.path_length > 0.0) ||
        (p.size_law != AUV_SIZES_UNIFORM && p.size_law != AUV_SIZES_BELL))
        return fail(h, AUV_ERR_ARG, "auv_generate_scene: parameters out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = pick(h, stream);
    h->gen_seed = seed;
    h->heading_stale = true;
    h->gen_params = p;
    HIP_TRY(h, hipMemsetAsync(bp<int32_t>(h, AUV_BUF_EPISODE), 0, (size_t)d.V * sizeof(int32_t), st));
    int rc = launch_scenegen(h, st, nullptr);
    if (rc) return rc;
    h->have_generator = true;
    h->have_scene = true;
    h->have_init = true;
    h->have_state = false;  // the live obstacle records follow at the next full reset
    return AUV_OK;
}

int auv_reset(auv_handle h, const uint8_t* mask, const double* init_state, void* stream) {
    if (!h) return AUV_ERR_ARG;
    if (!init_state && !h->have_init)
        return fail(h, AUV_ERR_ARG, "auv_reset: init_state is NULL and no initial states are on the device yet");
    int rc = ready_to_run(h, false);
    if (rc) return rc;
    if (mask && !h->have_state) return fail(h, AUV_ERR_STATE, "auv_reset: first reset must cover all envs (mask == NULL)");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = pick(h, stream);
    const Dims& d = h->d;
    if (init_state) {
        HIP_TRY(h, hipMemcpyAsync(h->init_stage, init_state, (size_t)d.V * 6 * sizeof(double), hipMemcpyHostToDevice, st));
        h->have_init = true;
    }
    if (mask) HIP_TRY(h, hipMemcpyAsync(h->mask_stage, mask, (size_t)d.V, hipMemcpyHostToDevice, st));
    ResetArgs ra;
    ra.d = d;
    ra.state = bp<double>(h, AUV_BUF_STATE);
    ra.t_step = bp<int32_t>(h, AUV_BUF_T_STEP);
    ra.path_seg = bp<int32_t>(h, AUV_BUF_PATH_SEG);
    ra.init = h->init_stage;
    ra.mask = mask ? h->mask_stage : nullptr;
    ra.circles = bp<double>(h, AUV_BUF_CIRCLES);
    ra.circles0 = h->circles0;
    ra.poly_pose = bp<double>(h, AUV_BUF_POLY_POSE);
    ra.poly_pose0 = h->poly_pose0;
    hipLaunchKernelGGL(k_reset, dim3((d.V + 255) / 256), dim3(256), 0, st, ra);
    HIP_TRY(h, hipGetLastError());
    h->have_state = true;
    rc = launch_sense(h, st, nullptr, 0);
    if (rc) return rc;
    // the staging copies read caller memory: finish before returning (a reset from the states already on the
    // device reads none and stays asynchronous, like a step)
    if (init_state || mask) HIP_TRY(h, hipStreamSynchronize(st));
    return AUV_OK;
}

int auv_step(auv_handle h, const double* action, void* stream) {
    if (!h) return AUV_ERR_ARG;
    int rc = ready_to_run(h, true);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    return do_step(h, action, pick(h, stream));
}

int auv_step_host(auv_handle h, const double* action, void* stream) {
    if (!h || !action) return fail(h, AUV_ERR_ARG, "auv_step_host: null argument");
    int rc = ready_to_run(h, true);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = pick(h, stream);
    HIP_TRY(h, hipMemcpyAsync(bp<double>(h, AUV_BUF_ACTION), action, (size_t)h->d.V * 2 * sizeof(double),
                              hipMemcpyHostToDevice, st));
    // pageable host memory: the copy above has consumed `action` on return
    return do_step(h, nullptr, st);
}

int auv_sense(auv_handle h, void* stream) {
    if (!h) return AUV_ERR_ARG;
    int rc = ready_to_run(h, true);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    return launch_sense(h, pick(h, stream), nullptr, 0);
}

int auv_get_buffer(auv_handle h, int which, void** dev_ptr, int64_t* shape, int32_t* ndim, int32_t* dtype) {
    if (!h) return AUV_ERR_ARG;
    if (which < 0 || which >= AUV_BUF_COUNT) return fail(h, AUV_ERR_ARG, "auv_get_buffer: unknown buffer %d", which);
    const BufDesc& b = h->buf[which];
    if (dev_ptr) *dev_ptr = b.ptr;
    if (shape)
        for (int i = 0; i < 4; ++i) shape[i] = i < b.ndim ? b.shape[i] : 0;
    if (ndim) *ndim = b.ndim;
    if (dtype) *dtype = b.dtype;
    return AUV_OK;
}

int auv_read_buffer(auv_handle h, int which, void* host_dst, int64_t nbytes, void* stream) {
    if (!h || !host_dst) return fail(h, AUV_ERR_ARG, "auv_read_buffer: null argument");
    if (which < 0 || which >= AUV_BUF_COUNT) return fail(h, AUV_ERR_ARG, "auv_read_buffer: unknown buffer %d", which);
    const BufDesc& b = h->buf[which];
    if (nbytes < 0 || (size_t)nbytes != b.bytes)
        return fail(h, AUV_ERR_ARG, "auv_read_buffer: buffer %d holds %zu bytes, caller asked for %lld", which,
                    b.bytes, (long long)nbytes);
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = pick(h, stream);
    if (b.bytes) HIP_TRY(h, hipMemcpyAsync(host_dst, b.ptr, b.bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return AUV_OK;
}

void* auv_host_alloc(int64_t nbytes) {
    void* p = nullptr;
    if (nbytes <= 0 || hipHostMalloc(&p, (size_t)nbytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void auv_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int auv_sync(auv_handle h, void* stream) {
    if (!h) return AUV_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(pick(h, stream)));
    return AUV_OK;
}

// DLPack (legacy DLManagedTensor ABI) -------------------------------------
namespace {
struct DlDevice { int32_t device_type; int32_t device_id; };
struct DlDataType { uint8_t code; uint8_t bits; uint16_t lanes; };
struct DlTensor {
    void* data;
    DlDevice device;
    int32_t ndim;
    DlDataType dtype;
    int64_t* shape;
    int64_t* strides;
    uint64_t byte_offset;
};
struct DlManagedTensor {
    DlTensor dl_tensor;
    void* manager_ctx;
    void (*deleter)(DlManagedTensor*);
};
struct DlExport {
    DlManagedTensor mt;
    int64_t shape[4];
};
void dl_delete(DlManagedTensor* mt) { delete reinterpret_cast<DlExport*>(mt); }
}  // namespace

void* auv_dlpack(auv_handle h, int which) {
    if (!h) return nullptr;
    if (which < 0 || which >= AUV_BUF_COUNT) {
        fail(h, AUV_ERR_ARG, "auv_dlpack: unknown buffer %d", which);
        return nullptr;
    }
    const BufDesc& b = h->buf[which];
    DlExport* e = new DlExport();
    static_assert(offsetof(DlExport, mt) == 0, "managed tensor must lead the export record");
    for (int i = 0; i < 4; ++i) e->shape[i] = b.shape[i];
    e->mt.dl_tensor.data = b.ptr;
    e->mt.dl_tensor.device = {10 /* kDLROCM */, h->device};
    e->mt.dl_tensor.ndim = b.ndim;
    switch (b.dtype) {
        case AUV_F64: e->mt.dl_tensor.dtype = {2, 64, 1}; break;
        case AUV_F32: e->mt.dl_tensor.dtype = {2, 32, 1}; break;
        case AUV_I32: e->mt.dl_tensor.dtype = {0, 32, 1}; break;
        case AUV_I64: e->mt.dl_tensor.dtype = {0, 64, 1}; break;
        default: e->mt.dl_tensor.dtype = {1, 8, 1}; break;
    }
    e->mt.dl_tensor.shape = e->shape;
    e->mt.dl_tensor.strides = nullptr;  // dense row-major
    e->mt.dl_tensor.byte_offset = 0;
    e->mt.manager_ctx = nullptr;
    e->mt.deleter = dl_delete;
    return &e->mt;
}

int auv_set_option(auv_handle h, int32_t key, int32_t value) {
    if (!h) return AUV_ERR_ARG;
    switch (key) {
        case AUV_OPT_LIDAR_MODE:
            if (value != AUV_LIDAR_RASTER && value != AUV_LIDAR_ALL_PAIRS)
                return fail(h, AUV_ERR_ARG, "auv_set_option: unknown LiDAR mode %d", value);
            if (value == AUV_LIDAR_ALL_PAIRS && h->auto_reset)
                return fail(h, AUV_ERR_ARG, "auv_set_option: the all-pairs path has no in-step restart; clear AUV_OPT_AUTO_RESET first");
            if (value == AUV_LIDAR_RASTER && h->custom_rays)
                return fail(h, AUV_ERR_ARG, "auv_set_option: the rasterising path needs the evenly spaced full-circle ray "
                            "table (ray i at -pi + (i+1) 2pi/R); the table set by auv_set_rays differs");
            if (value == AUV_LIDAR_RASTER && h->raster_lds > kLdsMax)
                return fail(h, AUV_ERR_ARG, "auv_set_option: rasterising LiDAR needs %zu B of LDS per env (limit 163840)",
                            h->raster_lds);
            h->lidar_mode = value;
            h->mode_forced_by_rays = false;  // the caller's own choice from here on
            return AUV_OK;
        case AUV_OPT_SENSE_BLOCK:
            if (value < 64 || value > 1024 || value % 64 != 0)
                return fail(h, AUV_ERR_ARG, "auv_set_option: k_sense workgroup size must be a multiple of 64, 64 to 1024");
            h->block_want = value;
            h->sense_block = value;
            return AUV_OK;
        case AUV_OPT_AUTO_RESET:
            if (value != 0 && value != 1) return fail(h, AUV_ERR_ARG, "auv_set_option: AUV_OPT_AUTO_RESET takes 0 or 1");
            if (value && h->lidar_mode != AUV_LIDAR_RASTER)
                return fail(h, AUV_ERR_ARG, "auv_set_option: in-step restart needs the rasterising path; this batch is on the "
                            "all-pairs path because %s", h->custom_rays ? "its ray table (auv_set_rays) is not the evenly spaced "
                            "full circle" : h->raster_lds > kLdsMax ? "one env's LDS image is too large for the rasterising kernel"
                            : "AUV_OPT_LIDAR_MODE selected it");
            h->auto_reset = value;
            return AUV_OK;
        case AUV_OPT_META_EVERY:
            if (value < 0) return fail(h, AUV_ERR_ARG, "auv_set_option: AUV_OPT_META_EVERY must be >= 0");
            h->meta_every = value;
            h->step_count = 0;
            return AUV_OK;
        case AUV_OPT_RISK_MODEL:
            if (value != AUV_RISK_CPA && value != AUV_RISK_REFERENCE)
                return fail(h, AUV_ERR_ARG, "auv_set_option: AUV_OPT_RISK_MODEL takes AUV_RISK_REFERENCE or AUV_RISK_CPA");
            h->risk_model = value;
            h->heading_stale = true;
            return AUV_OK;
        case AUV_OPT_REGENERATE:
            if (value != 0 && value != 1) return fail(h, AUV_ERR_ARG, "auv_set_option: AUV_OPT_REGENERATE takes 0 or 1");
            if (value && !h->have_generator)
                return fail(h, AUV_ERR_STATE, "auv_set_option: call auv_generate_scene before AUV_OPT_REGENERATE");
            h->regenerate = value;
            return AUV_OK;
        case AUV_OPT_RASTER_TILE:
            if (value < 0 || value > 64)
                return fail(h, AUV_ERR_ARG, "auv_set_option: envs per workgroup must be 0 (auto) or 1..64");
        {
            const int before = h->raster_want;
            set_raster_tile(h, value);
            if (h->raster_lds > kLdsMax) {
                const size_t need = h->raster_lds;
                set_raster_tile(h, before);
                return fail(h, AUV_ERR_ARG, "auv_set_option: a tile of %d envs needs %zu B of LDS (limit 163840)", value, need);
            }
            return AUV_OK;
        }
    }
    return fail(h, AUV_ERR_ARG, "auv_set_option: unknown key %d", key);
}

int auv_profile_enable(auv_handle h, int32_t on) {
    if (!h) return AUV_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    if (on && h->events.empty()) {
        h->events.resize((size_t)kProfRing * 4);
        for (auto& ev : h->events) HIP_TRY(h, hipEventCreate(&ev));
    }
    if (!on) {
        int rc = drain_profile(h);
        if (rc) return rc;
    }
    h->profiling = on != 0;
    return AUV_OK;
}

int auv_profile_read(auv_handle h, double ms[3], int64_t* n_steps) {
    if (!h || !ms || !n_steps) return fail(h, AUV_ERR_ARG, "auv_profile_read: null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = drain_profile(h);
    if (rc) return rc;
    for (int k = 0; k < 3; ++k) {
        ms[k] = h->prof_ms[k];
        h->prof_ms[k] = 0.0;
    }
    *n_steps = h->prof_steps;
    h->prof_steps = 0;
    return AUV_OK;
}

#ifdef AUV_DEBUG_BOUNDS
// debug builds only: violated index bounds counted by the kernels since the library was loaded
int auv_debug_violations(int64_t* out) {
    unsigned long long v = 0;
    if (!out || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_violations), sizeof v) != hipSuccess)
        return AUV_ERR_DEVICE;
    *out = (int64_t)v;
    return AUV_OK;
}
#endif

int auv_launch_info(auv_handle h, int32_t* out) {
    if (!h || !out) return AUV_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    const Dims& d = h->d;
    // the same choice of kernel and LDS size as the launch in sense_pass
    const bool cnn_inside = h->cnn_w && cnn_fits_sense_kernel(h->cnn_kind);
    const bool fixed = is_default_shape(d, h->raster_te, h->sense_block);
    const bool queued = !fixed && sense_wants_queue(h->raster_te, d.O, d.R);
    const size_t lds = fixed ? sense_layout(h->raster_te, d.R, d.C, d.P, d.NV, cnn_inside ? h->cnn_kind : -1, true).total
                             : h->raster_lds;
    const bool ref = h->risk_model == AUV_RISK_REFERENCE;
    const void* fn = fixed ? (ref ? reinterpret_cast<const void*>(k_sense_default<1>) : reinterpret_cast<const void*>(k_sense_default<0>))
                     : queued ? (ref ? reinterpret_cast<const void*>(k_sense_queued<1>) : reinterpret_cast<const void*>(k_sense_queued<0>))
                              : (ref ? reinterpret_cast<const void*>(k_sense<1>) : reinterpret_cast<const void*>(k_sense<0>));
    int per_cu = 0;
    if (lds <= kLdsMax) HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, h->sense_block, lds));
    out[0] = h->raster_te;
    out[1] = h->sense_block;
    out[2] = (int32_t)lds;
    out[3] = per_cu;
    return AUV_OK;
}

#ifdef AUV_PHASE_STAMPS
// diagnostic builds only: the launch shape of the sensing kernel {envs per workgroup, threads, LDS bytes, 0}
int auv_debug_geometry(auv_handle h, int32_t* out) {
    if (!h || !out) return AUV_ERR_ARG;
    out[0] = h->raster_te;
    out[1] = h->sense_block;
    out[2] = (int32_t)h->raster_lds;
    out[3] = 0;
    return AUV_OK;
}
// diagnostic builds only: device buffer [blocks * waves per block * 16] of i64 that k_sense stamps
int auv_debug_set_stamps(void* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof p) == hipSuccess ? AUV_OK : AUV_ERR_DEVICE;
}
#endif

// ---- episode statistics -------------------------------------------------------

int auv_episode_stats(auv_handle h, double* out, int32_t accumulate, void* stream) {
    if (!h) return AUV_ERR_ARG;
    if (!out) return fail(h, AUV_ERR_ARG, "auv_episode_stats: out is NULL");
    int rc = ready_to_run(h, true);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    return launch_stats(h, h->stats_ws, out, accumulate, pick(h, stream));
}

int auv_episode_stats_static(auv_handle h, double* out, int32_t accumulate, void* stream) {
    if (!h) return AUV_ERR_ARG;
    if (!out) return fail(h, AUV_ERR_ARG, "auv_episode_stats_static: out is NULL");
    if (h->d.V > AUV_STATS_MAX_ENVS)
        return fail(h, AUV_ERR_ARG, "auv_episode_stats_static: batch of %d envs exceeds the fixed scratch array (%d envs); "
                    "use auv_episode_stats", h->d.V, AUV_STATS_MAX_ENVS);
    int rc = ready_to_run(h, true);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    double* records = g_scratch_ptr[h->device];  // resolved by auv_create on this device
    if (!records) return fail(h, AUV_ERR_STATE, "auv_episode_stats_static: scratch array of device %d unresolved", h->device);
    return launch_stats(h, records, out, accumulate, pick(h, stream));
}

// ---- dataset rows assembled on the device ------------------------------------

int auv_dataset_rows(auv_handle h, int32_t phase, int32_t every, int32_t refresh_every, int32_t first, int32_t label_mode,
                     uint8_t* tracked, int64_t* row_of, int32_t* due_count, double* lidar_rows, double* risk_rows,
                     double* meta_rows, int32_t* risk_len, int32_t* meta_len, float* x_rows, float* label, void* stream) {
    if (!h) return AUV_ERR_ARG;
    if (every < 1 || refresh_every < 1) return fail(h, AUV_ERR_ARG, "auv_dataset_rows: every and refresh_every must be >= 1");
    if (phase != 0 && phase != 1) return fail(h, AUV_ERR_ARG, "auv_dataset_rows: phase must be 0 or 1");
    if (label_mode < ROWS_LABEL_NONE || label_mode > ROWS_LABEL_MEAN)
        return fail(h, AUV_ERR_ARG, "auv_dataset_rows: unknown label mode %d", label_mode);
    if (!due_count) return fail(h, AUV_ERR_ARG, "auv_dataset_rows: due_count is NULL");
    if (phase == 1 && (!tracked || !row_of || !lidar_rows || !risk_rows || !meta_rows || !risk_len || !meta_len))
        return fail(h, AUV_ERR_ARG, "auv_dataset_rows: null argument");
    if (phase == 1 && label_mode != ROWS_LABEL_NONE && (!x_rows || !label))
        return fail(h, AUV_ERR_ARG, "auv_dataset_rows: a label mode needs x_rows and label");
    int rc = ready_to_run(h, true);
    if (rc) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = pick(h, stream);
    const Dims& d = h->d;
    RowsArgs a;
    a.V = d.V;
    a.O = d.O;
    a.R = d.R;
    a.every = every;
    a.refresh_every = refresh_every;
    a.first = first ? 1 : 0;
    a.sensor_range = h->cfg.sensor_range;
    a.t_step = bp<int32_t>(h, AUV_BUF_T_STEP);
    a.meta = bp<double>(h, AUV_BUF_META);
    a.obst_dist = bp<double>(h, AUV_BUF_OBST_DIST);
    a.risk = bp<double>(h, AUV_BUF_RISK);
    a.own = bp<double>(h, AUV_BUF_OWN);
    a.ranges = bp<double>(h, AUV_BUF_RANGES);
    a.perception = bp<float>(h, AUV_BUF_PERCEPTION);
    a.tracked = tracked;
    a.row_of = reinterpret_cast<long long*>(row_of);
    a.due_count = due_count;
    a.lidar_rows = lidar_rows;
    a.risk_rows = risk_rows;
    a.meta_rows = meta_rows;
    a.risk_len = risk_len;
    a.meta_len = meta_len;
    a.x_rows = label_mode != ROWS_LABEL_NONE ? x_rows : nullptr;
    a.label = label_mode != ROWS_LABEL_NONE ? label : nullptr;
    a.label_mode = label_mode;
    if (phase == 0) {
        HIP_TRY(h, hipMemsetAsync(due_count, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_rows_due, dim3((unsigned)((d.V + 255) / 256)), dim3(256), 0, st, a);
    } else {
        const unsigned blocks = (unsigned)((d.V + 3) / 4);
        hipLaunchKernelGGL(k_rows_select, dim3(blocks), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_fmt_scan, dim3(1), dim3(1024), 0, st, a.row_of, (long long)d.V);
        hipLaunchKernelGGL(k_rows_write, dim3(blocks), dim3(256), 0, st, a);
    }
    HIP_TRY(h, hipGetLastError());
    return AUV_OK;
}

// ---- dataset text rows (host only) ------------------------------------------

int64_t auv_format_rows(const double* values, const int32_t* row_len, int64_t n_rows, int64_t stride, char* out,
                        int64_t out_cap) {
    if (!values || !row_len || !out || n_rows < 0 || stride < 0) return AUV_ERR_ARG;
    // upper bound of the text: 26 bytes per value + separator, 1 per row
    std::vector<int64_t> need((size_t)n_rows + 1, 0);
    for (int64_t i = 0; i < n_rows; ++i) {
        if (row_len[i] < 0 || row_len[i] > stride) return AUV_ERR_ARG;
        need[(size_t)i + 1] = need[(size_t)i] + (int64_t)row_len[i] * 27 + 1;
    }
    if (need[(size_t)n_rows] > out_cap) return AUV_ERR_MEMORY;
    // blocks of rows on the host threads, each into its worst-case slot of
    // `out`, then closed up in order
    int n_thr = (int)std::thread::hardware_concurrency();
    if (n_thr > 16) n_thr = 16;
    if (n_thr < 1 || need[(size_t)n_rows] < (1 << 20)) n_thr = 1;
    std::vector<int64_t> got((size_t)n_thr, 0), first((size_t)n_thr + 1, 0);
    for (int t = 0; t <= n_thr; ++t) first[(size_t)t] = n_rows * t / n_thr;
    auto work = [&](int t) {
        got[(size_t)t] = auvfmt::format_block(values, row_len, stride, first[(size_t)t], first[(size_t)t + 1],
                                              out + need[(size_t)first[(size_t)t]]);
    };
    if (n_thr == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_thr; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    int64_t pos = got[0];
    for (int t = 1; t < n_thr; ++t) {
        memmove(out + pos, out + need[(size_t)first[(size_t)t]], (size_t)got[(size_t)t]);
        pos += got[(size_t)t];
    }
    return pos;
}

int auv_format_rows_device(int device, const double* values, const int32_t* row_len, int64_t n_rows, int64_t stride,
                           char* text, int64_t text_cap, int64_t* row_off, int32_t* status, void* stream) {
    if (n_rows < 0 || n_rows > 0x7fffffffLL || stride < 1 || text_cap < 0) return AUV_ERR_ARG;
    if ((!values && n_rows > 0) || !text || !row_off || !status) return AUV_ERR_ARG;  // no rows: nothing is read
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return AUV_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return AUV_ERR_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(status, 0, 2 * sizeof(int32_t), st) != hipSuccess) return AUV_ERR_DEVICE;
    FmtArgs a;
    a.values = values;
    a.row_len = row_len;
    a.n_rows = n_rows;
    a.stride = stride;
    a.text = text;
    a.text_cap = text_cap;
    a.row_off = reinterpret_cast<long long*>(row_off);
    a.status = status;
    if (n_rows > 0) hipLaunchKernelGGL(k_fmt_row_bytes, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_fmt_scan, dim3(1), dim3(1024), 0, st, a.row_off, (long long)n_rows);
    if (n_rows > 0) hipLaunchKernelGGL(k_fmt_write, dim3((unsigned)n_rows), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? AUV_OK : AUV_ERR_DEVICE;
}

// ---- LiDAR-CNN consumers ----------------------------------------------------

struct auv_cnn {
    int kind = 0;
    int device = 0;
    float* w = nullptr;
    std::string err;
};

namespace {
thread_local std::string g_cnn_error;
int cnn_fail(auv_cnn* c, int code, const char* fmt, ...) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (c) c->err = tmp; else g_cnn_error = tmp;
    return code;
}
}  // namespace

const char* auv_cnn_last_error(auv_cnn_handle c) { return c ? c->err.c_str() : g_cnn_error.c_str(); }

int auv_cnn_n_weights(int32_t kind) {
    return (kind < AUV_CNN_SHALLOW || kind > AUV_CNN_DEEP) ? AUV_ERR_ARG : cnn_n_weights(kind);
}

int auv_cnn_out_dim(int32_t kind, int32_t head) {
    if (kind < AUV_CNN_SHALLOW || kind > AUV_CNN_DEEP || (head != AUV_CNN_FEATURES && head != AUV_CNN_RISK))
        return AUV_ERR_ARG;
    return cnn_out_dim(kind, head);
}

int auv_cnn_create(int32_t kind, const float* weights, int64_t n_weights, int device, auv_cnn_handle* out) {
    if (!weights || !out) return cnn_fail(nullptr, AUV_ERR_ARG, "auv_cnn_create: null argument");
    *out = nullptr;
    if (kind < AUV_CNN_SHALLOW || kind > AUV_CNN_DEEP)
        return cnn_fail(nullptr, AUV_ERR_ARG, "auv_cnn_create: unknown network %d", kind);
    if (n_weights != cnn_n_weights(kind))
        return cnn_fail(nullptr, AUV_ERR_ARG, "auv_cnn_create: network %d has %d parameters, got %lld", kind,
                        cnn_n_weights(kind), (long long)n_weights);
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return cnn_fail(nullptr, AUV_ERR_DEVICE, "auv_cnn_create: no HIP device (%s)",
                        e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n_dev) return cnn_fail(nullptr, AUV_ERR_ARG, "auv_cnn_create: device %d out of range", device);
    if (hipSetDevice(device) != hipSuccess) return cnn_fail(nullptr, AUV_ERR_DEVICE, "hipSetDevice failed");
    auv_cnn* c = new auv_cnn();
    c->kind = kind;
    c->device = device;
    // the deep network's copy also holds its two larger linear weights transposed (k_cnn_deep reads them by output)
    std::vector<float> host(weights, weights + n_weights);
    if (kind == CNN_DEEP) {
        host.resize(kDeepWeightsDev);
        for (int o = 0; o < 40; ++o)
            for (int k = 0; k < 72; ++k) host[3723 + (size_t)k * 40 + o] = weights[466 + o * 72 + k];
        for (int o = 0; o < 8; ++o)
            for (int k = 0; k < 40; ++k) host[3723 + kDeepL1T + (size_t)k * 8 + o] = weights[3386 + o * 40 + k];
    }
    if (hipMalloc((void**)&c->w, host.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(c->w, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        if (c->w) (void)hipFree(c->w);
        delete c;
        return cnn_fail(nullptr, AUV_ERR_MEMORY, "auv_cnn_create: weight upload failed");
    }
    *out = c;
    return AUV_OK;
}

void auv_cnn_destroy(auv_cnn_handle c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->w) (void)hipFree(c->w);
    delete c;
}

int auv_cnn_forward(auv_cnn_handle c, const float* scans, int64_t n_scans, int32_t head, float* out, void* stream) {
    if (!c) return AUV_ERR_ARG;
    if (!scans || !out) return cnn_fail(c, AUV_ERR_ARG, "auv_cnn_forward: null argument");
    if (head != AUV_CNN_FEATURES && head != AUV_CNN_RISK) return cnn_fail(c, AUV_ERR_ARG, "auv_cnn_forward: unknown head %d", head);
    if (n_scans < 0 || n_scans > ((int64_t)1 << 31) * kCnnScans / 2)
        return cnn_fail(c, AUV_ERR_ARG, "auv_cnn_forward: bad scan count %lld", (long long)n_scans);
    if (n_scans == 0) return AUV_OK;
    if (hipSetDevice(c->device) != hipSuccess) return cnn_fail(c, AUV_ERR_DEVICE, "hipSetDevice failed");
    CnnArgs a;
    a.scans = scans;
    a.w = c->w;
    a.out = out;
    a.B = n_scans;
    a.kind = c->kind;
    a.head = head;
    const unsigned blocks = (unsigned)((n_scans + kCnnScans - 1) / kCnnScans);
    if ((reinterpret_cast<uintptr_t>(scans) & 15u) != 0)
        return cnn_fail(c, AUV_ERR_ARG, "auv_cnn_forward: scans must be 16-byte aligned");
    if (a.kind == CNN_DEEP)
        hipLaunchKernelGGL(k_cnn_deep, dim3(blocks), dim3(256), cnn_deep_lds_bytes(), (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_cnn, dim3(blocks), dim3(256), cnn_lds_bytes(a.kind), (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cnn_fail(c, AUV_ERR_DEVICE, "k_cnn launch failed: %s", hipGetErrorString(e));
    return AUV_OK;
}

namespace {
// checks shared by the training entry points; fills the kernel's argument block
int train_args(const char* who, int device, const float* X, const float* y, int64_t n_rows, const int32_t* perm,
               int64_t n_batches, int32_t batch, float* weights, float* opt_state, const auv_train_params* hp,
               int64_t steps_done, float* batch_loss, TrainArgs* a) {
    if (!hp) return cnn_fail(nullptr, AUV_ERR_ARG, "%s: null argument", who);
    if (batch < 1 || batch > AUV_TRAIN_BATCH_MAX)
        return cnn_fail(nullptr, AUV_ERR_ARG, "%s: batch must be in 1..%d, got %d", who, AUV_TRAIN_BATCH_MAX, batch);
    if (n_rows < 1 || n_batches < 0 || steps_done < 0)
        return cnn_fail(nullptr, AUV_ERR_ARG, "%s: n_rows must be >= 1, n_batches and steps_done >= 0", who);
    if (hp->optimizer != AUV_OPT_ADAMW && hp->optimizer != AUV_OPT_SGD_MOMENTUM)
        return cnn_fail(nullptr, AUV_ERR_ARG, "%s: unknown optimizer %d", who, hp->optimizer);
    if (!(hp->lr > 0.0)) return cnn_fail(nullptr, AUV_ERR_ARG, "%s: lr must be positive", who);
    if (n_batches == 0) return AUV_OK;  // an epoch of fewer rows than one batch: nothing is read or written
    if (!X || !y || !perm || !weights || !opt_state || !batch_loss)
        return cnn_fail(nullptr, AUV_ERR_ARG, "%s: null argument", who);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return cnn_fail(nullptr, AUV_ERR_DEVICE, "%s: no HIP device %d", who, device);
    if (hipSetDevice(device) != hipSuccess) return cnn_fail(nullptr, AUV_ERR_DEVICE, "hipSetDevice failed");
    a->X = X;
    a->y = y;
    a->perm = perm;
    a->n_batches = n_batches;
    a->batch = batch;
    a->optimizer = hp->optimizer;
    a->step0 = steps_done;
    a->w = weights;
    a->opt = opt_state;
    a->batch_loss = batch_loss;
    a->lr = hp->lr;
    a->beta1 = hp->beta1;
    a->beta2 = hp->beta2;
    a->eps = hp->eps;
    a->weight_decay = hp->weight_decay;
    a->momentum = hp->momentum;
    return AUV_OK;
}

extern "C++" template <int KIND>
int launch_train_net(const TrainArgs& a, hipStream_t stream) {
    constexpr size_t lds = train_net_lds_bytes(KIND);
    static_assert(lds <= kLdsMax, "the batch's activations and gradients must fit the CU's LDS");
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_train_net<KIND>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return cnn_fail(nullptr, AUV_ERR_DEVICE, "k_train_net: %zu B of LDS refused: %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_train_net<KIND>, dim3(1), dim3(kTrainNetThreads), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cnn_fail(nullptr, AUV_ERR_DEVICE, "k_train_net launch failed: %s", hipGetErrorString(e));
    return AUV_OK;
}

// the two-launch form: a workgroup per sample, then the update; scratch from the stream-ordered allocator
extern "C++" template <int KIND>
int launch_train_split(const TrainArgs& t, const auv_train_params* hp, hipStream_t stream) {
    constexpr size_t lds = train_sample_lds_bytes(KIND);
    static_assert(lds <= 64 * 1024, "a sample's activations and the parameters fit the default LDS limit");
    constexpr int NPP = train_params_padded(KIND), NP = cnn_n_weights(KIND);
    float* scratch = nullptr;
    const size_t bytes = ((size_t)kTrainBatchMax * NPP + kTrainBatchMax) * sizeof(float);
    if (hipMallocAsync((void**)&scratch, bytes, stream) != hipSuccess)
        return cnn_fail(nullptr, AUV_ERR_MEMORY, "auv_train_cnn: %zu B of scratch refused", bytes);
    TrainSplitArgs a;
    a.t = t;
    a.gsample = scratch;
    a.err2 = scratch + (size_t)kTrainBatchMax * NPP;
    double pow1 = 1.0, pow2 = 1.0;  // running products, as the one-launch kernels form them
    for (long long s = 0; s < t.step0; ++s) {
        pow1 *= hp->beta1;
        pow2 *= hp->beta2;
    }
    for (long long b = 0; b < t.n_batches; ++b) {
        pow1 *= hp->beta1;
        pow2 *= hp->beta2;
        a.batch_index = b;
        a.pow1 = pow1;
        a.pow2 = pow2;
        hipLaunchKernelGGL(k_train_sample<KIND>, dim3((unsigned)t.batch), dim3(kTrainSampleThreads), lds, stream, a);
        hipLaunchKernelGGL(k_train_update<KIND>, dim3((NP + 255) / 256), dim3(256), 0, stream, a);
    }
    hipError_t e = hipGetLastError();
    (void)hipFreeAsync(scratch, stream);
    if (e != hipSuccess) return cnn_fail(nullptr, AUV_ERR_DEVICE, "k_train_sample launch failed: %s", hipGetErrorString(e));
    return AUV_OK;
}
}  // namespace

int auv_train_shallow(int device, const float* X, const float* y, int64_t n_rows, const int32_t* perm,
                      int64_t n_batches, int32_t batch, float* weights, float* opt_state,
                      const auv_train_params* hp, int64_t steps_done, float* batch_loss, void* stream) {
    TrainArgs a;
    const int rc = train_args("auv_train_shallow", device, X, y, n_rows, perm, n_batches, batch, weights, opt_state, hp,
                              steps_done, batch_loss, &a);
    if (rc != AUV_OK || n_batches == 0) return rc;
    hipLaunchKernelGGL(k_train_shallow, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cnn_fail(nullptr, AUV_ERR_DEVICE, "k_train_shallow launch failed: %s", hipGetErrorString(e));
    return AUV_OK;
}

int auv_train_cnn(int32_t kind, int device, const float* X, const float* y, int64_t n_rows, const int32_t* perm,
                  int64_t n_batches, int32_t batch, float* weights, float* opt_state, const auv_train_params* hp,
                  int64_t steps_done, float* batch_loss, void* stream) {
    if (kind < AUV_CNN_SHALLOW || kind > AUV_CNN_DEEP)
        return cnn_fail(nullptr, AUV_ERR_ARG, "auv_train_cnn: unknown network %d", kind);
    if (kind == AUV_CNN_SHALLOW)
        return auv_train_shallow(device, X, y, n_rows, perm, n_batches, batch, weights, opt_state, hp, steps_done,
                                 batch_loss, stream);
    TrainArgs a;
    const int rc = train_args("auv_train_cnn", device, X, y, n_rows, perm, n_batches, batch, weights, opt_state, hp,
                              steps_done, batch_loss, &a);
    if (rc != AUV_OK || n_batches == 0) return rc;
    return kind == AUV_CNN_2DEEP ? launch_train_net<CNN_2DEEP>(a, (hipStream_t)stream)
                                 : launch_train_split<CNN_DEEP>(a, hp, (hipStream_t)stream);
}

int auv_risk_rows(auv_handle h, int64_t n, const double* own_speed, const double* own_heading,
                  const double* heading_ts, const double* dist, const double* rel_speed, const double* bearing,
                  const double* course, double* out, void* stream) {
    if (!h) return AUV_ERR_ARG;
    if (n < 0 || n > ((int64_t)1 << 31) * 128) return fail(h, AUV_ERR_ARG, "auv_risk_rows: bad record count %lld", (long long)n);
    if (n == 0) return AUV_OK;
    if (!own_speed || !own_heading || !heading_ts || !dist || !rel_speed || !bearing || !course || !out)
        return fail(h, AUV_ERR_ARG, "auv_risk_rows: null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    RiskRowsArgs a;
    a.n = n;
    a.own_speed = own_speed;
    a.own_heading = own_heading;
    a.heading_ts = heading_ts;
    a.dist = dist;
    a.rel_speed = rel_speed;
    a.bearing = bearing;
    a.course = course;
    a.out = out;
    hipLaunchKernelGGL(k_risk_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(h, hipGetLastError());
    return AUV_OK;
}

int auv_attach_cnn(auv_handle h, auv_cnn_handle c, int32_t head) {
    if (!h) return AUV_ERR_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());  // no launch may still be using the old weights or tile
    BufDesc& b = h->buf[AUV_BUF_CNN_OUT];
    if (!c) {
        if (h->cnn_w) (void)hipFree(h->cnn_w);
        h->cnn_w = nullptr;
        set_raster_tile(h, h->raster_want);
        return AUV_OK;
    }
    if (head != AUV_CNN_FEATURES && head != AUV_CNN_RISK)
        return fail(h, AUV_ERR_ARG, "auv_attach_cnn: unknown head %d", head);
    if (h->d.R != kCnnLen)
        return fail(h, AUV_ERR_ARG, "auv_attach_cnn: the networks take %d-ray scans, this batch has %d", kCnnLen, h->d.R);
    if (c->device != h->device)
        return fail(h, AUV_ERR_ARG, "auv_attach_cnn: network lives on device %d, batch on %d", c->device, h->device);
    float* w = nullptr;
    // the deep network's device copy carries its transposed linear weights behind the parameters
    const size_t nb = (size_t)(c->kind == CNN_DEEP ? kDeepWeightsDev : cnn_n_weights(c->kind)) * sizeof(float);
    HIP_TRY(h, hipMalloc((void**)&w, nb));
    if (hipMemcpy(w, c->w, nb, hipMemcpyDeviceToDevice) != hipSuccess) {
        (void)hipFree(w);
        return fail(h, AUV_ERR_DEVICE, "auv_attach_cnn: weight copy failed");
    }
    float* old_w = h->cnn_w;
    const int old_kind = h->cnn_kind, old_head = h->cnn_head;
    h->cnn_w = w;
    h->cnn_kind = c->kind;
    h->cnn_head = head;
    set_raster_tile(h, h->raster_want);  // the activation buffers change the LDS image
    if (h->raster_lds > kLdsMax && h->lidar_mode == AUV_LIDAR_RASTER) {
        // the image no longer fits: refuse rather than change the sensing path (and lose the in-step restart) silently
        const size_t need = h->raster_lds;
        h->cnn_w = old_w;
        h->cnn_kind = old_kind;
        h->cnn_head = old_head;
        set_raster_tile(h, h->raster_want);
        (void)hipFree(w);
        return fail(h, AUV_ERR_ARG, "auv_attach_cnn: with this network in the sensing kernel one env needs %zu B of LDS "
                    "(limit 163840); select AUV_LIDAR_ALL_PAIRS first or run the network with auv_cnn_forward", need);
    }
    if (old_w) (void)hipFree(old_w);
    b.shape[1] = cnn_out_dim(c->kind, head);
    b.bytes = (size_t)b.shape[0] * (size_t)b.shape[1] * sizeof(float);
    return AUV_OK;
}

}  // extern "C"
