"""CPU oracle of the device-side scenario generator (TEST INFRASTRUCTURE ONLY).

SELF-DEFINED: the reference checkout holds no scenario generator (its
simulator package is absent), so the distributions below are this build's own
(static circles and constant-velocity ship polygons scattered round a vessel
at the origin, as in the dataset name "MovingObstaclesNoRules").  What IS
pinned is the random-number generator: Philox4x32-10 (Salmon et al., "Parallel
random numbers: as easy as 1, 2, 3", SC'11), checked against the published
known-answer vectors of the Random123 distribution in the tests.

Contract: every number is a function of (seed, env, episode, purpose, index)
only, so the device can regenerate one env's scene without looking at any
other env; doubles are built from 53 random bits exactly, and every
arithmetic step is unfused fp64 in the order written here (sin/cos are the
contract's `sincos`).
"""
from __future__ import annotations

import numpy as np

from oracle.auv_oracle import PI, TWO_PI, sincos

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

# purposes (third counter word)
K_VESSEL, K_CIRCLE_POS, K_CIRCLE_RAD, K_POLY_POS, K_POLY_MOTION, K_POLY_WIDTH, K_POLY_SPEED = range(7)

DEFAULT_PARAMS = dict(extent=200.0, r_lo=5.0, r_hi=30.0, w_lo=5.0, w_hi=20.0, s_lo=0.3, s_hi=1.2,
                      clearance=25.0, path_length=800.0, size_law=0)
PARAM_ORDER = ("extent", "r_lo", "r_hi", "w_lo", "w_hi", "s_lo", "s_hi", "clearance", "path_length")


def philox4x32_10(ctr, key):
    """ctr: 4 uint32 arrays (broadcastable), key: 2 uint32 scalars/arrays -> 4 uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & MASK32 for x in np.broadcast_arrays(*ctr)]
    k0 = np.uint64(int(key[0]) & 0xFFFFFFFF)
    k1 = np.uint64(int(key[1]) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & MASK32, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & MASK32]
        k0 = (k0 + np.uint64(W0)) & MASK32
        k1 = (k1 + np.uint64(W1)) & MASK32
    return [x.astype(np.uint32) for x in c]


def uniform2(seed, env, episode, purpose, index):
    """Two doubles in [0,1), each from 53 bits of one Philox block."""
    env = np.asarray(env)
    r = philox4x32_10((env, episode, np.full_like(env, purpose), index), (seed & 0xFFFFFFFF, seed >> 32))
    r = [x.astype(np.uint64) for x in r]
    m0 = (r[0] << np.uint64(21)) | (r[1] >> np.uint64(11))
    m1 = (r[2] << np.uint64(21)) | (r[3] >> np.uint64(11))
    s = 1.0 / 9007199254740992.0
    return m0.astype(np.float64) * s, m1.astype(np.float64) * s


def _push_out(x, y, clear):
    d = np.sqrt(x * x + y * y)
    d = np.where(d == 0.0, 1.0, d)
    f = np.where(d < clear, clear / d, 1.0)
    return x * f, y * f


def _draw_size(lo, hi, ua, ub, bell):
    """A size in [lo, hi]: uniform from ua, or (bell) a whole number lo..hi from the mean of two uniforms."""
    if bell:
        return np.floor(lo + ((hi + 1.0) - lo) * ((ua + ub) * 0.5))
    return lo + (hi - lo) * ua


def generate(seed, envs, episodes, C, P, NV, params=None):
    """Scene arrays of the given envs (1-D int arrays `envs`, `episodes`), laid
    out as gym_auv_amd.scenarios documents them."""
    p = dict(DEFAULT_PARAMS)
    p.update(params or {})
    envs = np.asarray(envs, dtype=np.int64)
    ep = np.broadcast_to(np.asarray(episodes, dtype=np.int64), envs.shape)
    n = envs.shape[0]
    E = p["extent"]
    u0, _ = uniform2(seed, envs, ep, K_VESSEL, np.zeros_like(envs))
    psi = u0 * TWO_PI - PI
    s, c = sincos(psi)
    init = np.zeros((6, n))
    init[2] = psi
    path = np.zeros((n, 6))
    path[:, 2], path[:, 3], path[:, 4], path[:, 5] = psi, c, s, p["path_length"]
    circles = np.zeros((n, C, 5))
    if C:
        e2, p2 = envs[:, None], ep[:, None]
        j = np.broadcast_to(np.arange(C), (n, C))
        ux, uy = uniform2(seed, e2, p2, K_CIRCLE_POS, j)
        ur, ur2 = uniform2(seed, e2, p2, K_CIRCLE_RAD, j)
        rad = _draw_size(p["r_lo"], p["r_hi"], ur, ur2, p["size_law"])
        cx, cy = _push_out(ux * (2.0 * E) - E, uy * (2.0 * E) - E, rad + p["clearance"])
        circles[..., 0], circles[..., 1], circles[..., 2] = cx, cy, rad
    pose = np.zeros((n, P, 6))
    verts = np.zeros((n, P, NV, 2))
    nv = np.zeros((n, P), dtype=np.int32)
    heading = np.zeros((n, P))
    if P:
        if NV < 5:
            raise ValueError("ship polygons need 5 vertex slots")
        e2, p2 = envs[:, None], ep[:, None]
        q = np.broadcast_to(np.arange(P), (n, P))
        ux, uy = uniform2(seed, e2, p2, K_POLY_POS, q)
        uh, us = uniform2(seed, e2, p2, K_POLY_MOTION, q)
        uw, uw2 = uniform2(seed, e2, p2, K_POLY_WIDTH, q)
        w = _draw_size(p["w_lo"], p["w_hi"], uw, uw2, p["size_law"])
        spd = p["s_lo"] + (p["s_hi"] - p["s_lo"]) * us
        if p["size_law"]:
            us1, us2 = uniform2(seed, e2, p2, K_POLY_SPEED, q)
            spd = p["s_lo"] + (p["s_hi"] - p["s_lo"]) * ((us1 + us2) * 0.5)
        h = uh * TWO_PI - PI
        X, Y = _push_out(ux * (2.0 * E) - E, uy * (2.0 * E) - E, 2.0 * w + p["clearance"])
        sh, ch = sincos(h)
        pose[..., 0], pose[..., 1], pose[..., 2], pose[..., 3] = X, Y, ch, sh
        pose[..., 4], pose[..., 5] = spd * ch, spd * sh
        hw = w * 0.5
        # ship outline: square stern, pointed bow, counter-clockwise
        verts[..., 0, 0], verts[..., 0, 1] = -w, -hw
        verts[..., 1, 0], verts[..., 1, 1] = 0.0, -hw
        verts[..., 2, 0], verts[..., 2, 1] = w, 0.0
        verts[..., 3, 0], verts[..., 3, 1] = 0.0, hw
        verts[..., 4, 0], verts[..., 4, 1] = -w, hw
        nv[:] = 5
        heading[:] = h
    return {"circles": circles, "poly_pose": pose, "poly_verts": verts, "poly_nv": nv, "poly_heading": heading,
            "init_state": init, "path": path}
