"""`AuvSim`: thin object wrapper over the C ABI (one handle = one batch of
envs on one GPU)."""
from __future__ import annotations

import atexit
import ctypes as C
import weakref

import numpy as np

from . import _lib, dlpack, params
from .params import ray_table

_NP_DTYPE = {_lib.F64: np.float64, _lib.F32: np.float32, _lib.I32: np.int32, _lib.U8: np.uint8,
             _lib.I64: np.int64}


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _stream_handle(stream):
    """Raw hipStream_t of a torch stream, an int, or None (HIP's default stream)."""
    if stream is None:
        return None
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)


_live = weakref.WeakSet()


@atexit.register
def _close_all():
    # handles must be destroyed while the HIP runtime is still up; at
    # interpreter exit __del__ would run after the runtime's own teardown
    for sim in list(_live):
        sim.close()


class AuvSim:
    def __init__(self, cfg: dict, device: int = 0, ray_dirs=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self._staging = {}
        self.cfg = dict(cfg)
        self.device = int(device)
        c = _lib.AuvConfig()
        c.abi_version = _lib.ABI_VERSION
        for k in ("n_envs", "n_rays", "n_circles", "n_polys", "n_verts", "max_steps"):
            setattr(c, k, int(cfg[k]))
        c.n_path_segments = int(cfg.get("n_path_segments", 1))
        for k in ("dt", "sensor_range", "vessel_radius", "thrust_max", "moment_max",
                  "look_ahead", "goal_radius"):
            setattr(c, k, float(cfg[k]))
        c.contact_dist = float(cfg.get("contact_dist", params.CONTACT_DIST_DEFAULT))
        c.risk[:] = [float(x) for x in cfg["risk"]]
        c.reward[:] = [float(x) for x in cfg["reward"]]
        model = np.asarray(cfg["model"], dtype=np.float64)
        if model.shape != (_lib.N_MODEL,):
            raise ValueError(f"cfg['model'] must hold {_lib.N_MODEL} numbers")
        c.model[:] = model.tolist()
        rc = self._lib.auv_create(C.byref(c), self.device, C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise _lib.AuvError(rc, self._lib.auv_last_error(None).decode())
        self.V, self.R = int(cfg["n_envs"]), int(cfg["n_rays"])
        self.C, self.P, self.NV = int(cfg["n_circles"]), int(cfg["n_polys"]), int(cfg["n_verts"])
        self.O = self.C + self.P
        _live.add(self)
        self.set_rays(ray_table(self.R) if ray_dirs is None else ray_dirs)
        self.set_option(_lib.OPT_RISK_MODEL, int(cfg.get("risk_model", _lib.RISK_REFERENCE)))

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            for p, _ in self._staging.values():
                self._lib.auv_host_free(p)
            self._staging = {}
            self._lib.auv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise _lib.AuvError(rc, self._lib.auv_last_error(self._h).decode())

    # -- inputs -----------------------------------------------------------
    def set_rays(self, ray_dirs):
        self.ray_dirs = _f64(ray_dirs, (self.R, 2))
        self._check(self._lib.auv_set_rays(self._h, _ptr(self.ray_dirs)))

    def set_scene(self, scene: dict, env_begin: int = 0):
        n = int(np.asarray(scene["circles"]).shape[0]) if self.C else int(np.asarray(scene["poly_pose"]).shape[0])
        circles = _f64(scene["circles"], (n, self.C, 5)) if self.C else None
        if self.P:
            pose = _f64(scene["poly_pose"], (n, self.P, 6))
            verts = _f64(scene["poly_verts"], (n, self.P, self.NV, 2))
            nv = np.ascontiguousarray(scene["poly_nv"], dtype=np.int32)
            if nv.shape != (n, self.P):
                raise ValueError("poly_nv shape")
            head = _f64(scene.get("poly_heading", np.zeros((n, self.P))), (n, self.P))
        else:
            pose = verts = nv = head = None
        path = scene.get("path")
        polyline = path is not None and np.asarray(path).ndim == 3
        straight = _f64(path, (n, 6)) if path is not None and not polyline else None
        self._check(self._lib.auv_set_scene(self._h, int(env_begin), n, _ptr(circles), _ptr(pose),
                                            _ptr(verts), _ptr(nv), _ptr(head), _ptr(straight)))
        if polyline:
            self.set_path(path, scene.get("path_nseg"), env_begin)

    def set_path(self, segments, n_segments=None, env_begin: int = 0):
        """Polyline paths [n,KS,6] = (x0, y0, angle, cos, sin, length) per segment
        (`scenarios.waypoint_path` makes them from waypoints), `n_segments` [n]
        in use per env (None: all); cfg["n_path_segments"] must cover KS."""
        seg = np.ascontiguousarray(segments, dtype=np.float64)
        if seg.ndim != 3 or seg.shape[2] != 6:
            raise ValueError("segments must be [n,KS,6]")
        cnt = None if n_segments is None else np.ascontiguousarray(n_segments, dtype=np.int32)
        if cnt is not None and cnt.shape != (seg.shape[0],):
            raise ValueError("n_segments shape")
        self._check(self._lib.auv_set_path(self._h, int(env_begin), seg.shape[0], _ptr(seg), seg.shape[1], _ptr(cnt)))

    def generate_scene(self, seed: int, params: dict | None = None, stream=None):
        """Scene, paths and initial states of every env from the device-side
        generator (episode 0); follow with `reset(None)`.  `params` overrides
        the defaults by name (extent, r_lo, r_hi, w_lo, w_hi, s_lo, s_hi,
        clearance, path_length, size_law; `scenarios.REFERENCE_LIKE` is the set that
        follows the sizes and speeds of the reference's logged rows)."""
        p = None
        if params is not None:
            full = dict(extent=200.0, r_lo=5.0, r_hi=30.0, w_lo=5.0, w_hi=20.0, s_lo=0.3, s_hi=1.2,
                        clearance=25.0, path_length=800.0, size_law=0)
            unknown = set(params) - set(full)
            if unknown:
                raise ValueError(f"unknown generator parameters {sorted(unknown)}")
            full.update(params)
            law = int(full.pop("size_law"))
            p = C.byref(_lib.AuvSceneParams(size_law=law, reserved=0, **{k: float(v) for k, v in full.items()}))
        self._check(self._lib.auv_generate_scene(self._h, int(seed) & (2 ** 64 - 1), p, _stream_handle(stream)))

    def reset(self, init_state, mask=None, stream=None):
        """init_state None: the initial states already on the device (generator or last reset)."""
        init = None if init_state is None else _f64(init_state, (6, self.V))
        m = None
        if mask is not None:
            m = np.ascontiguousarray(np.asarray(mask).astype(bool), dtype=np.uint8)
            if m.shape != (self.V,):
                raise ValueError("mask shape")
        self._check(self._lib.auv_reset(self._h, _ptr(m), _ptr(init), _stream_handle(stream)))

    # -- stepping ---------------------------------------------------------
    def step_host(self, action, stream=None):
        a = _f64(action, (2, self.V))
        self._check(self._lib.auv_step_host(self._h, _ptr(a), _stream_handle(stream)))

    def step_device(self, action_ptr, stream=None):
        """`action_ptr`: device address of f64 [2,V] (int), or None to reuse
        the library's action buffer."""
        p = None if action_ptr is None else C.c_void_p(int(action_ptr))
        self._check(self._lib.auv_step(self._h, p, _stream_handle(stream)))

    def sense(self, stream=None):
        self._check(self._lib.auv_sense(self._h, _stream_handle(stream)))

    def sync(self, stream=None):
        self._check(self._lib.auv_sync(self._h, _stream_handle(stream)))

    # -- outputs ----------------------------------------------------------
    def buffer(self, name):
        """(device pointer, shape tuple, auv_dtype) of a lent buffer."""
        which = _lib.BUFFER_NAMES[name] if isinstance(name, str) else int(name)
        p = C.c_void_p()
        shape = (C.c_int64 * 4)()
        nd = C.c_int32()
        dt = C.c_int32()
        self._check(self._lib.auv_get_buffer(self._h, which, C.byref(p), shape, C.byref(nd), C.byref(dt)))
        return p.value, tuple(shape[i] for i in range(nd.value)), dt.value

    def _pinned(self, which, shape, dtype):
        """Page-locked staging array for buffer `which` (kept until close)."""
        key = (which, tuple(shape))
        hit = self._staging.get(key)
        if hit is None:
            nbytes = max(1, int(np.prod(shape)) * np.dtype(dtype).itemsize)
            p = self._lib.auv_host_alloc(nbytes)
            if not p:
                raise MemoryError(f"cannot page-lock {nbytes} bytes")
            arr = np.frombuffer((C.c_char * nbytes).from_address(p), dtype=dtype, count=int(np.prod(shape)))
            hit = self._staging[key] = (p, arr.reshape(shape))
        return hit[1]

    def read(self, name, stream=None, pinned=False):
        """Host copy of a buffer as a numpy array (synchronises).  pinned=True:
        the copy lands in a page-locked staging array that belongs to this
        object and is overwritten by the next pinned read of the same buffer
        (about 3x the copy rate of pageable memory; for per-step readers)."""
        which = _lib.BUFFER_NAMES[name] if isinstance(name, str) else int(name)
        _, shape, dt = self.buffer(which)
        out = self._pinned(which, shape, _NP_DTYPE[dt]) if pinned else np.empty(shape, dtype=_NP_DTYPE[dt])
        self._check(self._lib.auv_read_buffer(self._h, which, _ptr(out), out.nbytes, _stream_handle(stream)))
        return out

    def tensor(self, name):
        """torch tensor aliasing a device buffer through DLPack (zero-copy)."""
        which = _lib.BUFFER_NAMES[name] if isinstance(name, str) else int(name)
        mt = self._lib.auv_dlpack(self._h, which)
        if not mt:
            raise _lib.AuvError(-1, self._lib.auv_last_error(self._h).decode())
        return dlpack.to_torch(mt)

    def set_option(self, key: int, value: int):
        self._check(self._lib.auv_set_option(self._h, int(key), int(value)))

    def attach_cnn(self, cnn, head: str = "features"):
        """Evaluate `head` of a `cnn.LidarCNN` inside every sensing pass; the
        result is the buffer "cnn_out" [V,out_dim].  None detaches."""
        from .cnn import HEADS
        self._check(self._lib.auv_attach_cnn(self._h, None if cnn is None else cnn._h, HEADS[head]))

    def risk_rows(self, own_speed, own_heading, heading_ts, dist, rel_speed, bearing, course, stream=None):
        """The reference's risk index (`RISK_REFERENCE`) of dataset records:
        float64 CUDA tensors of n elements each (speed_OS, heading_OS and the
        obstacle's heading_ts, dist, speed, bearing, course of a metadata row,
        utils/explore_dataset.py:8,40-53) -> float64 CUDA tensor [n], the value
        data/risk_*.csv holds for that record."""
        import torch
        cols = []
        for t in (own_speed, own_heading, heading_ts, dist, rel_speed, bearing, course):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.dim() == 1):
                raise TypeError("risk_rows takes 1-D float64 CUDA tensors")
            if t.device.index != self.device:
                raise ValueError(f"risk_rows: tensor on cuda:{t.device.index}, simulation on cuda:{self.device}")
            cols.append(t.contiguous())
        n = cols[0].numel()
        if any(t.numel() != n for t in cols):
            raise ValueError("risk_rows: columns differ in length")
        out = torch.empty(n, dtype=torch.float64, device=cols[0].device)
        if stream is None:
            stream = torch.cuda.current_stream(cols[0].device)
        self._check(self._lib.auv_risk_rows(self._h, n, *[C.c_void_p(t.data_ptr()) for t in cols],
                                            C.c_void_p(out.data_ptr()), _stream_handle(stream)))
        return out

    def episode_stats(self, out, accumulate: bool = False, stream=None):
        """Reduce the batch's current outputs into `out`, a float64 CUDA tensor of
        `_lib.STATS_FIELDS` elements (order: `dist.STAT_FIELDS`), on the device;
        accumulate=True adds to it."""
        import torch
        if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float64
                and out.is_contiguous() and out.numel() == _lib.STATS_FIELDS):
            raise TypeError(f"episode_stats takes a contiguous float64 CUDA tensor of {_lib.STATS_FIELDS} elements")
        if out.device.index != self.device:
            raise ValueError(f"episode_stats: tensor on cuda:{out.device.index}, simulation on cuda:{self.device}")
        if stream is None:
            stream = torch.cuda.current_stream(out.device)
        self._check(self._lib.auv_episode_stats(self._h, C.c_void_p(out.data_ptr()), 1 if accumulate else 0,
                                                _stream_handle(stream)))
        return out

    def launch_info(self) -> dict:
        """How the sensing kernel of this batch is launched (`auv_launch_info`)."""
        out = (C.c_int32 * 4)()
        self._check(self._lib.auv_launch_info(self._h, out))
        return {"envs_per_workgroup": out[0], "threads": out[1], "lds_bytes": out[2], "workgroups_per_cu": out[3]}

    # -- profiling ----------------------------------------------------------
    def profile(self, on: bool):
        self._check(self._lib.auv_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        """({'dynamics','encounter','lidar'} -> total ms, steps) since the last read."""
        ms = (C.c_double * 3)()
        n = C.c_int64()
        self._check(self._lib.auv_profile_read(self._h, ms, C.byref(n)))
        return {"dynamics": ms[0], "encounter": ms[1], "lidar": ms[2]}, int(n.value)
