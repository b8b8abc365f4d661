"""Simulation constants and the config record shared by host code and the
device library.

The vessel is a CyberShip-II-like 3-DOF surface craft.  The reference
checkout does not contain its ship model (only `radarCNN.py:232` names the
absent ``gym_auv`` package); the model numbers below are the ones that
reproduce the speed and heading columns of its logged dataset
(tests/golden/make_surge_golden.py, make_turn_golden.py).  The mounted files
fix the sensor suite directly: 180 rays
(`utils/dataloader.py:13`) of 150 m range (`utils/dataloader.py:56,127`).
"""
from __future__ import annotations

import numpy as np

N_RAYS_DEFAULT = 180          # utils/dataloader.py:13
SENSOR_RANGE_DEFAULT = 150.0  # utils/dataloader.py:56
NAV_FEATURES = 6              # radarCNN.py:185
META_FIELDS = 5               # utils/explore_dataset.py:45
N_MODEL = 16

# rigid body + added mass
_M = 23.8
_XG = 0.046
_IZ = 1.76
_XUD, _YVD, _YRD, _NVD, _NRD = -2.0, -10.0, 0.0, 0.0, -1.0


def model_constants() -> np.ndarray:
    """The 16 model numbers in the order the kernels index them:
    Minv11, Minv22, Minv23, Minv32, Minv33, m, m*x_g, 0 (reserved),
    Xu, Xuu, Yv, Yvv, Yr, Nv, Nr, Nrr (damping given as positive numbers).

    nu' = Minv (tau - N(nu) nu) with N(nu) nu = [Xu u, Yv v + (m u + Yr) r,
    Nv v + (m x_g u + Nr) r]: the linearised manoeuvring model (no v*r term in
    surge, rigid-body mass in the sway coupling).  These numbers reproduce the
    reference's logged speed and heading through its unactuated turns to
    1e-15 (tests/golden/make_turn_golden.py)."""
    m11 = _M - _XUD
    m22 = _M - _YVD
    m23 = _M * _XG - _YRD
    m32 = _M * _XG - _NVD
    m33 = _IZ - _NRD
    det = m22 * m33 - m23 * m32
    return np.array([
        1.0 / m11, m33 / det, -m23 / det, -m32 / det, m22 / det,
        _M, _M * _XG, 0.0,
        2.0, 0.0, 7.0, 0.0, 0.1, 0.1, 0.5, 0.0,
    ], dtype=np.float64)


def ray_table(n_rays: int = N_RAYS_DEFAULT) -> np.ndarray:
    """Body-frame unit directions [R,2] = (cos, sin) of ray i at angle
    -pi + (i+1)*2pi/R, i.e. the scan starts just past dead astern, runs
    counter-clockwise and ends dead astern, wrapping over the full circle as
    the circular padding of models/lidar_cnn_shallow.py:30 assumes."""
    i = np.arange(n_rays, dtype=np.float64)
    ang = -np.pi + (i + 1.0) * (2.0 * np.pi / n_rays)
    return np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang)], axis=1))


# Collision <=> the nearest obstacle's logged-style distance (hull distance less the 3.255 m the reference takes off
# every distance) falls below this: the hull within 1.255 m of the vessel's centre.  The reference's log rules out
# 0 (contact at 3.255 m): its vessel passed ships at hull distances down to 2.53 m in episodes that went on
# (tests/test_ref_contact_cpu.py).
CONTACT_DIST_DEFAULT = -2.0


def make_config(n_envs: int, n_circles: int, n_polys: int, n_verts: int = 5,
                n_rays: int = N_RAYS_DEFAULT, dt: float = 1.0,
                sensor_range: float = SENSOR_RANGE_DEFAULT,
                vessel_radius: float = 3.255, thrust_max: float = 2.0,
                moment_max: float = 0.075, look_ahead: float = 100.0,
                goal_radius: float = 5.0, max_steps: int = 10000,
                risk=(0.4, 0.6, 60.0), reward=(1.0, 100.0, 100.0, 0.01),
                contact_dist: float = CONTACT_DIST_DEFAULT) -> dict:
    """Plain dict understood by `AuvSim` (device) and by the CPU oracle."""
    return {
        "n_envs": int(n_envs), "n_rays": int(n_rays), "n_circles": int(n_circles),
        "n_polys": int(n_polys), "n_verts": int(n_verts), "dt": float(dt),
        "sensor_range": float(sensor_range), "vessel_radius": float(vessel_radius),
        "thrust_max": float(thrust_max), "moment_max": float(moment_max),
        "look_ahead": float(look_ahead), "goal_radius": float(goal_radius),
        "max_steps": int(max_steps), "risk": tuple(float(x) for x in risk),
        "reward": tuple(float(x) for x in reward),
        "model": model_constants(), "contact_dist": float(contact_dist),
    }
