/*
 * auvsim.h -- C ABI of libauvsim.so, the MI355X (gfx950) implementation of the
 * gym-auv per-step simulation hot path: fp64 3-DOF vessel dynamics, moving
 * obstacle advance, LiDAR ray-cast, signed obstacle distance / collision,
 * per-obstacle encounter metadata + risk, and observation assembly, vectorised
 * over V independent vessels (envs).
 *
 * The reference checkout (hebe2602/gym-auv @ v1) has no FFI or plugin layer
 * and does not contain the simulator itself; the interfaces below replace the
 * gym env contract as the mounted consumers see it:
 *   - observation dict {"perception": (B,1,180) closeness, "navigation":
 *     (B,1,6)}                                  radarCNN.py:165-175,184-193,215-262
 *   - closeness = 1 - range/150, 180 rays       utils/dataloader.py:13,56
 *   - metadata / risk dataset rows              utils/explore_dataset.py:40-53,
 *                                               utils/dataloader.py:104-120
 * Each entry point names the consumer-side line it serves.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative auv_status; no exception crosses the boundary.  One
 * handle per GPU per process; calls on one handle must be serialised by the
 * caller.  Device buffers are owned by the library and lent out by
 * auv_get_buffer() for the life of the handle.  Kernels are enqueued on the
 * hipStream_t given to the call (NULL = HIP's default stream, as in the HIP
 * API), so a PyTorch caller orders against them by passing its current
 * stream.
 */
#ifndef AUVSIM_H
#define AUVSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AUVSIM_ABI_VERSION 3   /* 2: auv_scene_params.size_law; 3: auv_config.contact_dist */
#define AUV_N_MODEL 16
#define AUV_NAV_FEATURES 6   /* radarCNN.py:185 */
#define AUV_META_FIELDS 5    /* utils/explore_dataset.py:45 */

typedef struct auv_sim* auv_handle;

typedef enum auv_status {
    AUV_OK = 0,
    AUV_ERR_ARG = -1,      /* bad argument / shape */
    AUV_ERR_DEVICE = -2,   /* no usable GPU, or a HIP call failed */
    AUV_ERR_MEMORY = -3,   /* device allocation failed */
    AUV_ERR_STATE = -4     /* call order (e.g. step before set_scene/reset) */
} auv_status;

/* Static description of one simulation batch.  Obstacle ids within an env:
 * circles 0..n_circles-1, then polygons n_circles..n_circles+n_polys-1. */
typedef struct auv_config {
    int32_t abi_version;    /* AUVSIM_ABI_VERSION */
    int32_t n_envs;         /* V  vessels, one per env */
    int32_t n_rays;         /* R  (180: utils/dataloader.py:13) */
    int32_t n_circles;      /* C  circle slots per env */
    int32_t n_polys;        /* P  polygon slots per env */
    int32_t n_verts;        /* NV vertex slots per polygon (3..16) */
    int32_t max_steps;      /* truncation horizon */
    int32_t n_path_segments; /* segment slots per path (0 or 1: straight paths; more: polylines, auv_set_path) */
    double dt;              /* integrator step, s */
    double sensor_range;    /* 150 m: utils/dataloader.py:56 */
    double vessel_radius;   /* clearance taken off every obstacle distance, m (3.255: what the reference's logged
                               dist column subtracts beside the obstacle's size) */
    double thrust_max;      /* surge force at action[0] = 1, N (2: the reference's logs saturate there) */
    double moment_max;      /* yaw moment at action[1] = +-1, N m (0.075: likewise) */
    double look_ahead;
    double goal_radius;
    double risk[3];         /* w_d, w_c, T_c of the AUV_RISK_CPA index */
    double reward[4];       /* w_progress, w_collision, w_goal, w_cross_track */
    double model[AUV_N_MODEL]; /* Minv11, Minv22, Minv23, Minv32, Minv33 (inverse inertia incl. added mass), m, m*x_g,
                                  reserved (0), Xu, Xuu, Yv, Yvv, Yr, Nv, Nr, Nrr (damping, positive):
                                  nu' = Minv (tau - [(Xu+Xuu|u|)u, (Yv+Yvv|v|)v + (m u + Yr) r,
                                  Nv v + (m x_g u + Nr + Nrr|r|) r]) */
    double contact_dist;    /* collision <=> the nearest obstacle's distance (AUV_BUF_MIN_DIST: hull distance less
                               vessel_radius) is below this.  Default -2.0: the hull within 1.255 m of the vessel's
                               centre.  0 (contact at vessel_radius itself) is contradicted by the reference's log,
                               whose vessel passed ships at hull distances down to 2.53 m in episodes that went on
                               (data/metadata_MovingObstaclesNoRules.csv rows 7362-7366, counting from 0) */
} auv_config;

typedef enum auv_dtype { AUV_F64 = 0, AUV_F32 = 1, AUV_I32 = 2, AUV_U8 = 3, AUV_I64 = 4 } auv_dtype;

/* Device buffers a caller may borrow (shape in elements). */
typedef enum auv_buffer {
    AUV_BUF_STATE = 0,       /* f64 [6,V]   x,y,psi,u,v,r (SoA) */
    AUV_BUF_ACTION = 1,      /* f64 [2,V]   staging for auv_step_host / action == NULL */
    AUV_BUF_RANGES = 2,      /* f64 [V,R]   metres, sensor_range when nothing hit */
    AUV_BUF_HIT = 3,         /* i32 [V,R]   obstacle id, -1 when nothing hit */
    AUV_BUF_PERCEPTION = 4,  /* f32 [V,1,R] obs["perception"], radarCNN.py:275 */
    AUV_BUF_NAVIGATION = 5,  /* f32 [V,1,6] obs["navigation"], radarCNN.py:190-193 */
    AUV_BUF_OBST_DIST = 6,   /* f64 [V,O]   signed hull distance less vessel_radius, +inf if absent */
    AUV_BUF_RISK = 7,        /* f64 [V,O]   risk_*.csv values, utils/dataloader.py:108 */
    AUV_BUF_META = 8,        /* f64 [V,O,5] utils/explore_dataset.py:45 */
    AUV_BUF_OWN = 9,         /* f64 [V,2]   speed_OS, heading_OS, utils/explore_dataset.py:8 */
    AUV_BUF_COLLISION = 10,  /* u8  [V] */
    AUV_BUF_REACHED_GOAL = 11, /* u8 [V] */
    AUV_BUF_TRUNCATED = 12,  /* u8  [V] */
    AUV_BUF_DONE = 13,       /* u8  [V] */
    AUV_BUF_MIN_DIST = 14,   /* f64 [V] */
    AUV_BUF_MIN_IDX = 15,    /* i32 [V] */
    AUV_BUF_ALONG = 16,      /* f64 [V]   along-track position on the path */
    AUV_BUF_T_STEP = 17,     /* i32 [V] */
    AUV_BUF_CIRCLES = 18,    /* f64 [V,C,5] cx,cy,rad,vx,vy.  The live obstacle records: read them freely, but change a
                                scene through auv_set_scene / auv_generate_scene only: the library keeps per-scene
                                constants derived from them (headings, beams, turned vertices, the initial poses) */
    AUV_BUF_POLY_POSE = 19,  /* f64 [V,P,6] X,Y,cos h,sin h,vx,vy (same rule) */
    AUV_BUF_REWARD = 20,     /* f64 [V]   reward of the last step (0 after reset) */
    AUV_BUF_CNN_OUT = 21,    /* f32 [V,out_dim] output of the attached network (auv_attach_cnn); zeros, out_dim 12, before */
    AUV_BUF_EPISODE = 22,    /* i32 [V]   in-step restarts of the env so far (= its episode number for the generator) */
    AUV_BUF_PATH_SEG = 23,   /* i32 [V]   the segment of its path each vessel is on */
    AUV_BUF_COUNT = 24
} auv_buffer;

/* Tunables (auv_set_option).  Both LiDAR kernels produce identical bits; the
 * all-pairs one tests every ray against every primitive and exists as the
 * cross-check of the rasterising default. */
typedef enum auv_option {
    AUV_OPT_LIDAR_MODE = 0,   /* AUV_LIDAR_RASTER (default) | AUV_LIDAR_ALL_PAIRS */
    AUV_OPT_RASTER_TILE = 1,  /* envs per workgroup of the rasterising kernel; 0 = auto */
    AUV_OPT_SENSE_BLOCK = 2,  /* threads per k_sense workgroup: a multiple of 64, 64 to 1024 (default by the
                               * primitives, circles + edges, of a tile: 128 up to 64 of them, 256 up to 511,
                               * 512 up to 1023, 768 beyond) */
    AUV_OPT_AUTO_RESET = 3,   /* 1: an env whose done flag is set restarts INSIDE the next auv_step, with no
                               * host round trip: its action is ignored, the vessel takes the initial state
                               * of the most recent auv_reset call, its step counter is cleared, its
                               * obstacles return to their uploaded poses, its reward is 0 and its flags
                               * are those of the fresh state (rasterising path only; default 0) */
    AUV_OPT_REGENERATE = 4,   /* 1: with AUV_OPT_AUTO_RESET, a restarting env also gets a NEW scene, path and
                               * initial state from the generator (auv_generate_scene must have been called),
                               * for episode number AUV_BUF_EPISODE + 1; default 0: same scene every episode */
    AUV_OPT_META_EVERY = 5,   /* n: the per-obstacle bearing, course (AUV_BUF_META fields 3, 4) and AUV_BUF_RISK
                               * are refreshed only by every n-th auv_step since this call (and by every
                               * auv_reset / auv_sense); in between they keep their last values.  The
                               * reference logs them about every 5th step (dataset rows); a policy does not
                               * read them.  0: never during steps.  Default 1: every step.  Distances,
                               * collision, reward and the observations are always current.  Rasterising
                               * path only (the all-pairs path always refreshes). */
    AUV_OPT_RISK_MODEL = 6    /* which index fills AUV_BUF_RISK.  AUV_RISK_REFERENCE (default): the reference's own
                               * index as recovered from data/metadata_MovingObstaclesNoRules.csv against
                               * data/risk_MovingObstaclesNoRules.csv (0.3 DCPA x TCPA + 0.3 range + 0.2 bearing
                               * + 0.2 target-aspect memberships, clipped at 0, in the reference's frames); it
                               * reproduces all 23,680 per-obstacle records of those files to 2e-15.  With it
                               * the heading_ts, speed and course fields of AUV_BUF_META follow the reference's
                               * conventions too (heading_ts = direction of the obstacle's velocity, 0 at rest;
                               * course = world-frame direction of the relative velocity as the reference forms
                               * it, 0 / -pi for an obstacle at rest; refreshed with bearing and risk, see
                               * AUV_OPT_META_EVERY), so that auv_risk_rows of a metadata row gives back the
                               * AUV_BUF_RISK row, as in the reference's files.  AUV_RISK_CPA: a plain
                               * closest-point-of-approach index weighted by auv_config.risk, with heading_ts =
                               * the uploaded polygon heading, speed = |v_obst - v_own|, course relative to
                               * the vessel's heading. */
} auv_option;
#define AUV_LIDAR_RASTER 0
#define AUV_LIDAR_ALL_PAIRS 1
#define AUV_RISK_CPA 0
#define AUV_RISK_REFERENCE 1

/* Library build information: "libauvsim <abi> gfx950 ...". */
const char* auv_version(void);

/* Last error text of a handle, or (h == NULL) of the calling thread's last
 * failed auv_create. */
const char* auv_last_error(auv_handle h);

/* Create a batch on HIP device `device`.  Replaces constructing the env
 * (consumer: PerceptionNavigationExtractor.__init__, radarCNN.py:202-267,
 * which reads the observation space this config implies). */
int auv_create(const auv_config* cfg, int device, auv_handle* out);
void auv_destroy(auv_handle h);

/* Body-frame ray directions, host f64 [R,2] = (cos, sin) per ray.  The
 * rasterising path (AUV_LIDAR_RASTER, the default) indexes rays by angle and
 * needs the evenly spaced full circle, ray i at -pi + (i + 1) 2 pi / R (what
 * the reference's 180-ray scan is pinned to, to +-2 rays, by its logged rows
 * and shipped networks); a table that differs from it by more than 1e-9
 * switches the batch to AUV_LIDAR_ALL_PAIRS, which reads the directions only,
 * and AUV_LIDAR_RASTER is then refused by auv_set_option.  Uploading the
 * standard table again puts a batch that was switched this way back on
 * AUV_LIDAR_RASTER; a path selected through AUV_OPT_LIDAR_MODE is left alone. */
int auv_set_rays(auv_handle h, const double* ray_dirs);

/* Upload scene data for envs [env_begin, env_begin+env_count), front to back
 * (a call may start anywhere inside the part already uploaded); stepping needs
 * every env covered.  All host
 * pointers: circles f64 [n,C,5], poly_pose f64 [n,P,6], poly_verts f64
 * [n,P,NV,2], poly_nv i32 [n,P], poly_heading f64 [n,P], path f64 [n,6].
 * A pointer may be NULL when its slot count is 0 (path NULL keeps paths). */
int auv_set_scene(auv_handle h, int32_t env_begin, int32_t env_count,
                  const double* circles, const double* poly_pose,
                  const double* poly_verts, const int32_t* poly_nv,
                  const double* poly_heading, const double* path);

/* Piecewise-straight paths for envs [env_begin, env_begin+env_count): host f64
 * segments [n, n_slots, 6] = x0, y0, angle, cos angle, sin angle, length per
 * segment (n_slots <= auv_config.n_path_segments), host i32 n_segments [n]
 * segments in use per env or NULL (= n_slots).  A vessel is on one segment
 * (AUV_BUF_PATH_SEG; the first after a reset) and moves on to the next when its
 * projection passes the segment's end, never back.  The navigation features
 * (the 6 numbers of obs["navigation"], radarCNN.py:184-193) are then: along-track
 * position = arc length before the segment + projection on it; cross-track and
 * heading errors of the current segment; look-ahead heading error towards the
 * point look_ahead further along the path.  The goal is the end of the last
 * segment.  auv_set_scene's `path` is the one-segment case. */
int auv_set_path(auv_handle h, int32_t env_begin, int32_t env_count, const double* segments, int32_t n_slots,
                 const int32_t* n_segments);

/* Parameters of the scenario generator (metres, m/s). */
typedef struct auv_scene_params {
    double extent;       /* obstacle centres ~ U(-extent, extent)^2 round the vessel at the origin */
    double r_lo, r_hi;   /* circle radius ~ U */
    double w_lo, w_hi;   /* ship beam ~ U (the hull is 2 beams long) */
    double s_lo, s_hi;   /* ship speed ~ U, heading ~ U(-pi, pi) */
    double clearance;    /* no hull starts nearer than this to the vessel */
    double path_length;  /* straight path from the origin along the vessel's initial heading */
    int32_t size_law;    /* AUV_SIZES_UNIFORM (0): radius, beam and speed uniform over [lo, hi].
                            AUV_SIZES_BELL (1): radius and beam are whole numbers lo..hi with a triangular law
                            (floor of lo + (hi + 1 - lo)(u + u')/2), speed triangular over [lo, hi]: the shape of
                            the sizes and speeds recovered from the reference's logged rows (whole radii 15..41 m
                            peaking near 29, whole ship sizes 1..18 m peaking near 8, speeds 0.06..1.2 m/s
                            peaking near 0.5) */
    int32_t reserved;
} auv_scene_params;
#define AUV_SIZES_UNIFORM 0
#define AUV_SIZES_BELL 1

/* Generate scene, paths and initial states of EVERY env on the device, for
 * episode 0 (clears AUV_BUF_EPISODE): `n_circles` static circles and `n_polys`
 * five-vertex ships per env (needs n_verts >= 5 when n_polys > 0), each number
 * a function of (seed, env, episode, purpose, index) through Philox4x32-10.
 * params == NULL: extent 200, radius 5..30, beam 5..20, speed 0.3..1.2,
 * clearance 25, path 800, uniform sizes.  Afterwards auv_reset may be called with
 * init_state == NULL.  Replaces the scenario constructor of the env
 * (MovingObstaclesNoRules, the scenario named by the reference's data files
 * data/ *_MovingObstaclesNoRules.csv).  Asynchronous on `stream`. */
int auv_generate_scene(auv_handle h, uint64_t seed, const auv_scene_params* params, void* stream);

/* gym reset(): host init_state f64 [6,V] (NULL: the initial states of the last
 * reset call or of the generator); mask u8 [V] or NULL (= all envs).
 * Resets the selected envs (vessel state, step counter, obstacles back to
 * their uploaded or generated poses), then senses.
 * Observation consumer: radarCNN.py:270-280. */
int auv_reset(auv_handle h, const uint8_t* mask, const double* init_state, void* stream);

/* gym step(): action is a DEVICE pointer f64 [2,V] (surge in [0,1], steer in
 * [-1,1]) or NULL to use AUV_BUF_ACTION.  Advances obstacles, integrates the
 * vessels, then senses.  Asynchronous on `stream`. */
int auv_step(auv_handle h, const double* action, void* stream);

/* Same with a HOST action pointer (copied to AUV_BUF_ACTION first). */
int auv_step_host(auv_handle h, const double* action, void* stream);

/* Recompute every output from the current state without advancing time. */
int auv_sense(auv_handle h, void* stream);

/* Borrow a device buffer: pointer, shape (up to 4 dims), ndim, auv_dtype. */
int auv_get_buffer(auv_handle h, int which, void** dev_ptr, int64_t* shape,
                   int32_t* ndim, int32_t* dtype);

/* Copy a buffer to host memory (synchronises `stream`). */
int auv_read_buffer(auv_handle h, int which, void* host_dst, int64_t nbytes, void* stream);

/* Page-locked host memory for auv_read_buffer / auv_step_host / auv_reset: copies
 * to and from it run at the full PCIe rate (pageable memory is staged by the
 * runtime at roughly a third of it).  NULL when the allocation fails.  No
 * counterpart in the reference (its env lives in host memory). */
void* auv_host_alloc(int64_t nbytes);
void auv_host_free(void* p);

/* Borrow a device buffer as a DLPack DLManagedTensor* (device type kDLROCM),
 * malloc'ed by the library; the consumer's call of its deleter frees the
 * record, never the buffer.  A Python caller wraps the pointer in a PyCapsule
 * named "dltensor" for torch.from_dlpack (consumer: the tensors that
 * radarCNN.py:275 feeds to the CNN).  NULL on error. */
void* auv_dlpack(auv_handle h, int which);

/* Block until all work enqueued on `stream` is done. */
int auv_sync(auv_handle h, void* stream);

/* Set a tunable; takes effect from the next reset/step/sense. */
int auv_set_option(auv_handle h, int32_t key, int32_t value);

/* Per-kernel device timing.  While enabled, auv_step brackets each kernel
 * with HIP events on the launch stream.  auv_profile_read synchronises and
 * returns the accumulated milliseconds per kernel {dynamics, encounter,
 * lidar} and the number of steps they cover, then clears the accumulators. */
int auv_profile_enable(auv_handle h, int32_t on);
int auv_profile_read(auv_handle h, double ms[3], int64_t* n_steps);

/* (no counterpart in the reference) How the sensing kernel of this batch is launched on the rasterising path, as it
 * stands (scene shape, options, attached network): out = {envs per workgroup, threads per workgroup, LDS bytes per
 * workgroup, workgroups of it that one compute unit holds at a time (the runtime's occupancy figure for that
 * kernel, block size and LDS size)}.  For benchmarks and tuning: launches nothing. */
int auv_launch_info(auv_handle h, int32_t out[4]);

/* ---- episode statistics, reduced on the device ------------------------------
 * out (DEVICE f64 [AUV_STATS_FIELDS]) = over the batch's current outputs:
 * {number of envs, sum of AUV_BUF_DONE, of AUV_BUF_COLLISION, of
 * AUV_BUF_REACHED_GOAL, of AUV_BUF_TRUNCATED, of AUV_BUF_REWARD, sum over envs of
 * the nearest LiDAR return (min over the rays of AUV_BUF_RANGES), sum over envs
 * of the largest obstacle risk (max over AUV_BUF_RISK, 0 without obstacles)}.
 * accumulate != 0 adds to `out` instead of overwriting it, so a rollout's
 * statistics build up on the device and cross to the host (or into the
 * all-reduce of a multi-GPU job) once.  Two small launches on `stream`, a
 * workspace owned by the handle, sums folded in a fixed order (the same bits
 * on every call).  The task-level numbers the reference reports per agent
 * (results_RL/DRL_performance.txt:1-10: progress, collision avoidance, time) are
 * ratios of these sums. */
#define AUV_STATS_FIELDS 8
int auv_episode_stats(auv_handle h, double* out, int32_t accumulate, void* stream);

/* auv_episode_stats for callers that cannot allocate: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no
 * counterpart in the reference.  Same arguments, same kernels, same bits in
 * `out`; the per-workgroup records go to a fixed array in the library's own
 * device image (1 MiB per GPU the library runs on, held until the process ends)
 * instead of the handle's workspace, and the call allocates and frees nothing.
 * Limits: batches of at most AUV_STATS_MAX_ENVS envs (larger ones fail with
 * AUV_ERR_ARG and a message); every call on one GPU shares that GPU's array,
 * so calls that overlap on one GPU (different streams or handles) race -- order
 * them on one stream.  The array keeps the last call's records; a call reads
 * back only the records it has just written, so nothing has to be cleared. */
#define AUV_STATS_MAX_ENVS 1048576
int auv_episode_stats_static(auv_handle h, double* out, int32_t accumulate, void* stream);

/* ---- dataset rows assembled on the device -------------------------------------
 * What the reference's logger keeps per logged sample (the rows of LiDAR_*, risk_* and metadata_*.csv:
 * utils/dataloader.py:55,59,108, utils/explore_dataset.py:8,40-53), from the batch's buffers, without the host:
 * phase 0 counts the envs whose episode is at a multiple of `every` steps into due_count[0] (the obstacle
 * lists are only touched on steps where some env is due: read it back, and stop there if it is 0);
 * phase 1 brings every env's obstacle list up to date in `tracked` [V,O] u8 (rebuilt at multiples of
 * `refresh_every` steps of the episode, or everywhere when `first` != 0, from the obstacles whose logged
 * distance is below the sensor range; in between a list only loses obstacles whose slot went away), and
 * writes one row per due env with a non-empty list, in env order: lidar_rows [n,R]; risk_rows [n,O] and
 * meta_rows [n,2+5*O] with the listed obstacles first, id order kept; risk_len, meta_len [n]; row_of [V+1]
 * i64 = the row of every env, row_of[V] = n.  With label_mode AUV_ROWS_LABEL_MAX / SUM / MEAN also the
 * sample the reference's loader makes of the row (utils/dataloader.py:44-120): x_rows [n,R] f32 closeness
 * and label [n] f32 over at most 8 of the row's risks.  All pointers are DEVICE memory sized for V rows.
 * Feed the three row arrays to auv_format_rows_device for the files' text.  Asynchronous on `stream`. */
#define AUV_ROWS_LABEL_NONE 0
#define AUV_ROWS_LABEL_MAX 1
#define AUV_ROWS_LABEL_SUM 2
#define AUV_ROWS_LABEL_MEAN 3
int auv_dataset_rows(auv_handle h, int32_t phase, int32_t every, int32_t refresh_every, int32_t first, int32_t label_mode,
                     uint8_t* tracked, int64_t* row_of, int32_t* due_count, double* lidar_rows, double* risk_rows,
                     double* meta_rows, int32_t* risk_len, int32_t* meta_len, float* x_rows, float* label, void* stream);

/* ---- dataset text rows (host only, no GPU involved) -------------------------
 * Formats n_rows ragged rows of doubles as the reference's dataset files do
 * (numpy.savetxt defaults: "%.18e", single spaces, "\n"; the files read by
 * utils/dataloader.py:55,59,108 and utils/explore_dataset.py:39).  Row i holds
 * row_len[i] values starting at values[i*stride].  Returns the number of bytes
 * written to `out`, AUV_ERR_MEMORY if out_cap is too small (27 bytes per value
 * + 1 per row always suffice), AUV_ERR_ARG on a null pointer. */
int64_t auv_format_rows(const double* values, const int32_t* row_len, int64_t n_rows, int64_t stride,
                        char* out, int64_t out_cap);
/* The same text formed on the device (the rows leave the GPU as bytes ready for the file; the digits are the
 * host formatter's arithmetic in 192-bit integers).  All pointers are DEVICE memory on `device`:
 *   values [n_rows, stride] f64, row_len [n_rows] i32 or NULL (every row holds `stride` values),
 *   text [text_cap] bytes out (26 per value + 1 per row always suffice),
 *   row_off [n_rows + 1] i64 out: where each row starts in `text`; row_off[n_rows] = bytes in all,
 *   status [2] i32 out: [0] values the device does not format (it covers zero and 1e-36 <= |v| < 1e18:
 *   everything a dataset row holds; infinities, NaN and magnitudes outside are counted, and the caller
 *   formats that block with auv_format_rows instead), [1] rows that did not fit text_cap (not written).
 * Asynchronous on `stream`; read status and row_off[n_rows] after synchronising. */
int auv_format_rows_device(int device, const double* values, const int32_t* row_len, int64_t n_rows, int64_t stride,
                           char* text, int64_t text_cap, int64_t* row_off, int32_t* status, void* stream);

/* ---- dataset records -> risk (device pointers) -----------------------------
 * The AUV_RISK_REFERENCE index evaluated on n dataset records: per obstacle
 * record of a metadata row (utils/explore_dataset.py:8,40-53) the vessel's
 * speed_OS and heading_OS and the obstacle's heading_ts, dist, speed, bearing
 * and course; out[i] is the value the reference logged for that record in
 * data/risk_*.csv (utils/dataloader.py:108 reads those files).  All pointers
 * are f64 device arrays of n elements on the handle's GPU.  Asynchronous on
 * `stream`. */
int auv_risk_rows(auv_handle h, int64_t n, const double* own_speed, const double* own_heading,
                  const double* heading_ts, const double* dist, const double* rel_speed, const double* bearing,
                  const double* course, double* out, void* stream);

/* ---- LiDAR-CNN consumers of the scan (fp32, 180-ray scans) -----------------
 * Forward passes of the three reference networks, reading the perception
 * buffer in place.  Weights: the network's state dict, tensors flattened and
 * concatenated in state-dict order (59 / 178 / 3,723 floats).
 *   AUV_CNN_SHALLOW  models/lidar_cnn_shallow.py:23-52; features head =
 *                    LidarCNN_shallow_pretrained, radarCNN.py:165-182 (12)
 *   AUV_CNN_2DEEP    models/lidar_cnn_deep.py:110-157; features head =
 *                    LidarCNN_not_so_deep_pretrained, radarCNN.py:110-146 (12)
 *   AUV_CNN_DEEP     models/lidar_cnn_deep.py:19-96 with channels [2,4,4,6],
 *                    kernel 9 (trainer.py:144); features head =
 *                    LidarCNN_deep_pretrained, radarCNN.py:25-94 (8)
 * AUV_CNN_RISK is the full network of models/ (one risk value per scan). */
typedef struct auv_cnn* auv_cnn_handle;
#define AUV_CNN_SHALLOW 0
#define AUV_CNN_2DEEP 1
#define AUV_CNN_DEEP 2
#define AUV_CNN_FEATURES 0
#define AUV_CNN_RISK 1

int auv_cnn_n_weights(int32_t kind);              /* parameter count, or AUV_ERR_ARG */
int auv_cnn_out_dim(int32_t kind, int32_t head);  /* outputs per scan, or AUV_ERR_ARG */
int auv_cnn_create(int32_t kind, const float* weights /* host */, int64_t n_weights, int device,
                   auv_cnn_handle* out);
void auv_cnn_destroy(auv_cnn_handle c);
/* scans: DEVICE f32 [n_scans,1,180]; out: DEVICE f32 [n_scans,out_dim].  Asynchronous on `stream`. */
int auv_cnn_forward(auv_cnn_handle c, const float* scans, int64_t n_scans, int32_t head, float* out,
                    void* stream);
const char* auv_cnn_last_error(auv_cnn_handle c);

/* ---- training the one-convolution network on the device ---------------------
 * Trainer.train_batch (trainer.py:44-53: forward, MSE loss, backward, optimizer
 * step, zero_grad) over every batch of an epoch (trainer.py:80-83) in ONE
 * launch, for LidarCNN_shallow (models/lidar_cnn_shallow.py:23-52).  All
 * pointers but `hp` are DEVICE memory on `device`:
 *   X [n_rows,180] f32 scans (closeness), y [n_rows] f32 labels,
 *   perm [n_batches*batch] i32 row index of every slot of every batch (the
 *   DataLoader's shuffle, utils/dataloader.py:76-80; batch <= 16); every entry
 *   must lie in [0, n_rows): the kernels index X and y with it unchecked (it is
 *   device memory; trainer.DeviceTrainer checks it on the host before the upload),
 *   weights [59] f32 in/out (state-dict order, as auv_cnn_create takes them),
 *   opt_state [118] f32 in/out (AdamW: exp_avg, exp_avg_sq; SGD: momentum
 *   buffer; zeros before the first step), batch_loss [n_batches] f32 out (what
 *   train_batch returns per batch).  steps_done = optimizer steps before this
 *   call (AdamW's bias correction).  The update formulas are torch.optim's
 *   (AdamW: trainer.py:33, SGD momentum: trainer.py:35).  Asynchronous on
 *   `stream`. */
typedef struct auv_train_params {
    int32_t optimizer;     /* AUV_OPT_ADAMW | AUV_OPT_SGD_MOMENTUM */
    int32_t reserved;
    double lr;
    double beta1, beta2, eps, weight_decay;  /* AdamW (torch defaults 0.9, 0.999, 1e-8, 0.01) */
    double momentum;                         /* SGD (trainer.py:35: 0.9) */
} auv_train_params;
#define AUV_OPT_ADAMW 0
#define AUV_OPT_SGD_MOMENTUM 1
#define AUV_TRAIN_BATCH_MAX 16
int auv_train_shallow(int device, const float* X, const float* y, int64_t n_rows, const int32_t* perm,
                      int64_t n_batches, int32_t batch, float* weights, float* opt_state,
                      const auv_train_params* hp, int64_t steps_done, float* batch_loss, void* stream);
/* The same call for any of the three networks (Trainer.train takes whichever model it is given,
 * trainer.py:18-43,144-151): kind = AUV_CNN_SHALLOW (59 parameters), AUV_CNN_2DEEP
 * (models/lidar_cnn_deep.py:110-146; 178) or AUV_CNN_DEEP (models/lidar_cnn_deep.py:19-82 with channels
 * [2,4,4,6], kernel 9; 3,723); weights [n] and opt_state [2n] for the network's n parameters, in the order
 * auv_cnn_create takes them.  The two small networks: one workgroup keeps the parameters, the batch's
 * activations and their gradients in LDS and runs every batch of the epoch in sequence, in one launch.
 * AUV_CNN_DEEP: two launches per batch, enqueued for the whole epoch by this call (a workgroup per sample for
 * forward and backward, then the optimizer step); its 233 KiB of scratch come from and return to the
 * stream-ordered allocator (hipMallocAsync / hipFreeAsync on `stream`). */
int auv_train_cnn(int32_t kind, int device, const float* X, const float* y, int64_t n_rows, const int32_t* perm,
                  int64_t n_batches, int32_t batch, float* weights, float* opt_state, const auv_train_params* hp,
                  int64_t steps_done, float* batch_loss, void* stream);

/* Attach a network to a batch: from then on every sensing pass (auv_reset,
 * auv_step*, auv_sense) also evaluates `head` of the network on the fresh
 * scans and leaves the result in AUV_BUF_CNN_OUT, f32 [V,out_dim].  On the
 * rasterising path this happens inside the sensing kernel, on the closeness
 * values it still holds on chip; results equal auv_cnn_forward on
 * AUV_BUF_PERCEPTION bit for bit.  The batch keeps its own copy of the weights
 * (`c` may be destroyed afterwards); c == NULL detaches.  Needs n_rays == 180.
 * Synchronises the device.  Replaces the extractor's per-step CNN call,
 * radarCNN.py:270-280. */
int auv_attach_cnn(auv_handle h, auv_cnn_handle c, int32_t head);

#ifdef __cplusplus
}
#endif
#endif /* AUVSIM_H */
