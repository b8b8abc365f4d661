"""ctypes binding of libauvsim.so (declarations mirror include/auvsim.h).

There is no CPU fallback: if the shared library has not been built, or no
GPU is usable, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AUVSIM_LIB points diagnostic runs at another build of the same library
LIB_PATH = os.environ.get("AUVSIM_LIB") or os.path.join(_HERE, "libauvsim.so")

ABI_VERSION = 3
N_MODEL = 16

# auv_dtype
F64, F32, I32, U8, I64 = 0, 1, 2, 3, 4

# auv_buffer
(BUF_STATE, BUF_ACTION, BUF_RANGES, BUF_HIT, BUF_PERCEPTION, BUF_NAVIGATION,
 BUF_OBST_DIST, BUF_RISK, BUF_META, BUF_OWN, BUF_COLLISION, BUF_REACHED_GOAL,
 BUF_TRUNCATED, BUF_DONE, BUF_MIN_DIST, BUF_MIN_IDX, BUF_ALONG, BUF_T_STEP,
 BUF_CIRCLES, BUF_POLY_POSE, BUF_REWARD, BUF_CNN_OUT, BUF_EPISODE, BUF_PATH_SEG, BUF_COUNT) = range(25)

BUFFER_NAMES = {
    "state": BUF_STATE, "action": BUF_ACTION, "ranges": BUF_RANGES, "hit": BUF_HIT,
    "perception": BUF_PERCEPTION, "navigation": BUF_NAVIGATION, "obst_dist": BUF_OBST_DIST,
    "risk": BUF_RISK, "meta": BUF_META, "own": BUF_OWN, "collision": BUF_COLLISION,
    "reached_goal": BUF_REACHED_GOAL, "truncated": BUF_TRUNCATED, "done": BUF_DONE,
    "min_dist": BUF_MIN_DIST, "min_idx": BUF_MIN_IDX, "along": BUF_ALONG,
    "t_step": BUF_T_STEP, "circles": BUF_CIRCLES, "poly_pose": BUF_POLY_POSE,
    "reward": BUF_REWARD, "cnn_out": BUF_CNN_OUT, "episode": BUF_EPISODE, "path_seg": BUF_PATH_SEG,
}

# auv_option
OPT_LIDAR_MODE, OPT_RASTER_TILE, OPT_SENSE_BLOCK, OPT_AUTO_RESET, OPT_REGENERATE, OPT_META_EVERY = 0, 1, 2, 3, 4, 5
OPT_RISK_MODEL = 6
LIDAR_RASTER, LIDAR_ALL_PAIRS = 0, 1
RISK_CPA, RISK_REFERENCE = 0, 1
STATS_FIELDS = 8
SIZES_UNIFORM, SIZES_BELL = 0, 1
OPT_ADAMW, OPT_SGD_MOMENTUM = 0, 1
ROWS_LABEL = {None: 0, "max": 1, "sum": 2, "mean": 3}
TRAIN_BATCH_MAX = 16
STATS_MAX_ENVS = 1048576

# CNN consumers
CNN_SHALLOW, CNN_2DEEP, CNN_DEEP = 0, 1, 2
CNN_FEATURES, CNN_RISK = 0, 1

# every symbol include/auvsim.h declares
EXPORTS = (
    "auv_version", "auv_last_error", "auv_create", "auv_destroy", "auv_set_rays",
    "auv_set_scene", "auv_reset", "auv_step", "auv_step_host", "auv_sense",
    "auv_get_buffer", "auv_read_buffer", "auv_dlpack", "auv_sync", "auv_set_option",
    "auv_profile_enable", "auv_cnn_n_weights", "auv_cnn_out_dim", "auv_cnn_create", "auv_cnn_destroy",
    "auv_cnn_forward", "auv_cnn_last_error", "auv_format_rows", "auv_format_rows_device", "auv_dataset_rows",
    "auv_profile_read", "auv_attach_cnn", "auv_generate_scene", "auv_host_alloc", "auv_host_free",
    "auv_risk_rows", "auv_episode_stats", "auv_episode_stats_static", "auv_train_shallow", "auv_train_cnn", "auv_set_path",
    "auv_launch_info",
)


class AuvConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_envs", C.c_int32), ("n_rays", C.c_int32),
        ("n_circles", C.c_int32), ("n_polys", C.c_int32), ("n_verts", C.c_int32),
        ("max_steps", C.c_int32), ("n_path_segments", C.c_int32),
        ("dt", C.c_double), ("sensor_range", C.c_double), ("vessel_radius", C.c_double),
        ("thrust_max", C.c_double), ("moment_max", C.c_double), ("look_ahead", C.c_double),
        ("goal_radius", C.c_double), ("risk", C.c_double * 3), ("reward", C.c_double * 4),
        ("model", C.c_double * N_MODEL), ("contact_dist", C.c_double),
    ]


class AuvSceneParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("extent", "r_lo", "r_hi", "w_lo", "w_hi", "s_lo", "s_hi", "clearance",
                                          "path_length")] + [("size_law", C.c_int32), ("reserved", C.c_int32)]


class AuvTrainParams(C.Structure):
    _fields_ = [("optimizer", C.c_int32), ("reserved", C.c_int32), ("lr", C.c_double), ("beta1", C.c_double),
                ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double), ("momentum", C.c_double)]


class AuvError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libauvsim error {code}: {message}")
        self.code = code


_lib = None


def load():
    """Load libauvsim.so once; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; a process must run
    # ONE HIP runtime, or the second one finds no GPU and device pointers are
    # not shared.  Importing torch first makes its runtime the one that
    # libauvsim's NEEDED libamdhip64.so.7 resolves to (same soname), so the
    # DLPack views and torch streams belong to the runtime the kernels use.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    dp = C.POINTER(C.c_double)
    lib.auv_version.restype = C.c_char_p
    lib.auv_version.argtypes = []
    lib.auv_last_error.restype = C.c_char_p
    lib.auv_last_error.argtypes = [vp]
    lib.auv_create.restype = C.c_int
    lib.auv_create.argtypes = [C.POINTER(AuvConfig), C.c_int, C.POINTER(vp)]
    lib.auv_destroy.restype = None
    lib.auv_destroy.argtypes = [vp]
    lib.auv_set_rays.restype = C.c_int
    lib.auv_set_rays.argtypes = [vp, vp]
    lib.auv_set_scene.restype = C.c_int
    lib.auv_set_scene.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.auv_reset.restype = C.c_int
    lib.auv_reset.argtypes = [vp, vp, vp, vp]
    lib.auv_step.restype = C.c_int
    lib.auv_step.argtypes = [vp, vp, vp]
    lib.auv_step_host.restype = C.c_int
    lib.auv_step_host.argtypes = [vp, vp, vp]
    lib.auv_sense.restype = C.c_int
    lib.auv_sense.argtypes = [vp, vp]
    lib.auv_get_buffer.restype = C.c_int
    lib.auv_get_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(i64), C.POINTER(i32),
                                   C.POINTER(i32)]
    lib.auv_read_buffer.restype = C.c_int
    lib.auv_read_buffer.argtypes = [vp, C.c_int, vp, i64, vp]
    lib.auv_dlpack.restype = vp
    lib.auv_dlpack.argtypes = [vp, C.c_int]
    lib.auv_set_option.restype = C.c_int
    lib.auv_set_option.argtypes = [vp, i32, i32]
    lib.auv_sync.restype = C.c_int
    lib.auv_sync.argtypes = [vp, vp]
    lib.auv_profile_enable.restype = C.c_int
    lib.auv_profile_enable.argtypes = [vp, i32]
    lib.auv_profile_read.restype = C.c_int
    lib.auv_profile_read.argtypes = [vp, dp, C.POINTER(i64)]
    lib.auv_launch_info.restype = C.c_int
    lib.auv_launch_info.argtypes = [vp, C.POINTER(i32)]
    lib.auv_format_rows.restype = i64
    lib.auv_format_rows.argtypes = [vp, vp, i64, i64, vp, i64]
    lib.auv_dataset_rows.restype = C.c_int
    lib.auv_dataset_rows.argtypes = [vp, i32, i32, i32, i32, i32] + [vp] * 11
    lib.auv_format_rows_device.restype = C.c_int
    lib.auv_format_rows_device.argtypes = [C.c_int, vp, vp, i64, i64, vp, i64, vp, vp, vp]
    lib.auv_cnn_n_weights.restype = C.c_int
    lib.auv_cnn_n_weights.argtypes = [i32]
    lib.auv_cnn_out_dim.restype = C.c_int
    lib.auv_cnn_out_dim.argtypes = [i32, i32]
    lib.auv_cnn_create.restype = C.c_int
    lib.auv_cnn_create.argtypes = [i32, vp, i64, C.c_int, C.POINTER(vp)]
    lib.auv_cnn_destroy.restype = None
    lib.auv_cnn_destroy.argtypes = [vp]
    lib.auv_cnn_forward.restype = C.c_int
    lib.auv_cnn_forward.argtypes = [vp, vp, i64, i32, vp, vp]
    lib.auv_cnn_last_error.restype = C.c_char_p
    lib.auv_cnn_last_error.argtypes = [vp]
    lib.auv_generate_scene.restype = C.c_int
    lib.auv_generate_scene.argtypes = [vp, C.c_uint64, C.POINTER(AuvSceneParams), vp]
    lib.auv_host_alloc.restype = vp
    lib.auv_host_alloc.argtypes = [i64]
    lib.auv_host_free.restype = None
    lib.auv_host_free.argtypes = [vp]
    lib.auv_attach_cnn.restype = C.c_int
    lib.auv_attach_cnn.argtypes = [vp, vp, i32]
    lib.auv_risk_rows.restype = C.c_int
    lib.auv_risk_rows.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.auv_episode_stats.restype = C.c_int
    lib.auv_episode_stats.argtypes = [vp, vp, i32, vp]
    lib.auv_episode_stats_static.restype = C.c_int
    lib.auv_episode_stats_static.argtypes = [vp, vp, i32, vp]
    lib.auv_train_shallow.restype = C.c_int
    lib.auv_train_shallow.argtypes = [C.c_int, vp, vp, i64, vp, i64, i32, vp, vp, C.POINTER(AuvTrainParams), i64, vp, vp]
    lib.auv_train_cnn.restype = C.c_int
    lib.auv_train_cnn.argtypes = [i32] + lib.auv_train_shallow.argtypes
    lib.auv_set_path.restype = C.c_int
    lib.auv_set_path.argtypes = [vp, i32, i32, vp, i32, vp]
    _lib = lib
    return lib
