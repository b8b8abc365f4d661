"""ctypes front of the C restatement of the oracle (TEST INFRASTRUCTURE ONLY).

`COracleEnv` has the interface and buffers of `auv_oracle.OracleEnv` (without
the in-step restart) and produces the same bits; env ranges can be stepped by
several threads (ctypes drops the GIL during the call).  Build:
`make -C oracle` (done by `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libauvoracle.so")


class OcConfig(C.Structure):
    _fields_ = [("V", C.c_int32), ("R", C.c_int32), ("C", C.c_int32), ("P", C.c_int32), ("NV", C.c_int32),
                ("max_steps", C.c_int32), ("risk_model", C.c_int32), ("KS", C.c_int32),
                ("dt", C.c_double), ("sensor_range", C.c_double), ("vessel_radius", C.c_double),
                ("thrust_max", C.c_double), ("moment_max", C.c_double), ("look_ahead", C.c_double),
                ("goal_radius", C.c_double), ("risk", C.c_double * 3), ("reward", C.c_double * 4),
                ("model", C.c_double * 16), ("contact_dist", C.c_double)]


_BUFS = ("state", "t_step", "circles", "poly_pose", "poly_verts", "poly_heading", "path", "ray_dirs", "poly_nv",
         "along", "ranges", "hit", "perception", "obst_dist", "collision", "min_dist", "min_idx", "own", "meta",
         "risk", "navigation", "reached_goal", "truncated", "done", "reward", "path_nseg", "path_seg")


class OcBuffers(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in _BUFS]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            # the checker builds itself where a C compiler is at hand (seconds)
            import subprocess

            try:
                subprocess.run(["make", "-C", os.path.dirname(os.path.abspath(__file__)), "-s"], check=True)
            except (OSError, subprocess.CalledProcessError) as e:
                raise ImportError(f"{LIB_PATH} is missing and `make -C oracle` failed: {e}")
        _lib = C.CDLL(LIB_PATH)
        _lib.oc_run.restype = None
        _lib.oc_run.argtypes = [C.POINTER(OcConfig), C.POINTER(OcBuffers), C.c_void_p, C.c_int32, C.c_int32]
    return _lib


class COracleEnv:
    def __init__(self, cfg, ray_dirs, threads: int = 1):
        self.lib = load()
        self.cfg = cfg
        V, R = cfg["n_envs"], cfg["n_rays"]
        Cn, P, NV = cfg["n_circles"], cfg["n_polys"], cfg["n_verts"] if cfg["n_polys"] else 1
        O = Cn + P
        c = OcConfig()
        c.V, c.R, c.C, c.P, c.NV, c.max_steps = V, R, Cn, P, NV, cfg["max_steps"]
        c.risk_model = int(cfg.get("risk_model", 1))
        for k in ("dt", "sensor_range", "vessel_radius", "thrust_max", "moment_max", "look_ahead", "goal_radius"):
            setattr(c, k, float(cfg[k]))
        c.risk[:] = [float(x) for x in cfg["risk"]]
        c.reward[:] = [float(x) for x in cfg["reward"]]
        c.model[:] = [float(x) for x in cfg["model"]]
        c.contact_dist = float(cfg.get("contact_dist", -2.0))
        self._c = c
        self.threads = max(1, int(threads))
        self._pool = ThreadPoolExecutor(self.threads) if self.threads > 1 else None
        f8, i4, u1, f4 = np.float64, np.int32, np.uint8, np.float32
        self.ray_dirs = np.ascontiguousarray(ray_dirs, dtype=f8)
        self.state = np.zeros((6, V), f8)
        self.t_step = np.zeros(V, i4)
        straight = np.zeros((V, 6), f8)
        straight[:, 3] = 1.0
        self.set_path(straight)
        self.circles = np.zeros((V, Cn, 5), f8)
        self.poly_pose = np.zeros((V, P, 6), f8)
        self.poly_verts = np.zeros((V, P, NV, 2), f8)
        self.poly_nv = np.zeros((V, P), i4)
        self.poly_heading = np.zeros((V, P), f8)
        self.along = np.zeros(V, f8)
        self.ranges = np.zeros((V, R), f8)
        self.hit = np.zeros((V, R), i4)
        self.perception = np.zeros((V, 1, R), f4)
        self.obst_dist = np.zeros((V, O), f8)
        self.collision = np.zeros(V, u1)
        self.min_dist = np.zeros(V, f8)
        self.min_idx = np.zeros(V, i4)
        self.own = np.zeros((V, 2), f8)
        self.meta = np.zeros((V, O, 5), f8)
        self.risk = np.zeros((V, O), f8)
        self.navigation = np.zeros((V, 1, 6), f4)
        self.reached_goal = np.zeros(V, u1)
        self.truncated = np.zeros(V, u1)
        self.done = np.zeros(V, u1)
        self.reward = np.zeros(V, f8)

    def _buffers(self):
        b = OcBuffers()
        for k in _BUFS:
            a = getattr(self, k)
            assert a.flags["C_CONTIGUOUS"]
            setattr(b, k, a.ctypes.data)
        return b

    def _run(self, action):
        V = self.cfg["n_envs"]
        b = self._buffers()
        ap = None if action is None else action.ctypes.data_as(C.c_void_p)
        if self._pool is None:
            self.lib.oc_run(C.byref(self._c), C.byref(b), ap, 0, V)
            return
        n = self.threads
        cuts = [V * i // n for i in range(n + 1)]
        list(self._pool.map(lambda i: self.lib.oc_run(C.byref(self._c), C.byref(b), ap, cuts[i], cuts[i + 1]),
                            range(n)))

    def set_scene(self, circles, poly_pose, poly_verts, poly_nv, poly_heading=None):
        self.circles[...] = circles
        self.poly_pose[...] = poly_pose
        if self.cfg["n_polys"]:
            self.poly_verts[...] = poly_verts
            self.poly_nv[...] = poly_nv
            if poly_heading is not None:
                self.poly_heading[...] = poly_heading
        self.circles0 = self.circles.copy()
        self.poly_pose0 = self.poly_pose.copy()

    def set_path(self, path, nseg=None):
        """One straight segment per env [V,6], or a polyline [V,KS,6] with `nseg` [V] segments in use; the
        library reads the table [V,KS,8] = (x0, y0, angle, cos, sin, length, arc length before, 0)."""
        path = np.asarray(path, dtype=np.float64)
        if path.ndim == 2:
            path = path[:, None, :]
        V, KS = path.shape[:2]
        tab = np.zeros((V, KS, 8))
        tab[:, :, :6] = path
        cum = np.zeros(V)
        for k in range(KS):
            tab[:, k, 6] = cum
            cum = cum + path[:, k, 5]
        self.path = np.ascontiguousarray(tab)
        self.path_nseg = np.full(V, KS, dtype=np.int32) if nseg is None else np.ascontiguousarray(nseg, dtype=np.int32)
        self.path_seg = np.zeros(V, dtype=np.int32)
        self._c.KS = KS

    def reset(self, init_state, mask=None):
        init_state = np.asarray(init_state, dtype=np.float64)
        if mask is None:
            m = slice(None)
        else:
            m = np.asarray(mask).astype(bool)
        self.state[:, m] = init_state[:, m]
        self.t_step[m] = 0
        self.path_seg[m] = 0
        self.circles[m] = self.circles0[m]
        self.poly_pose[m] = self.poly_pose0[m]
        self._run(None)

    def step(self, action):
        self._run(np.ascontiguousarray(action, dtype=np.float64))

    def close(self):
        if self._pool is not None:
            self._pool.shutdown()
