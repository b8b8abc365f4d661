"""Vessel dynamics against the reference's own data.

The reference's simulator is not in its checkout, but its dataset logs
[speed_OS, heading_OS] once per 5 simulator steps
(data/metadata_MovingObstaclesNoRules.csv; utils/explore_dataset.py:8).  Between
consecutive rows with bit-identical headings the vessel ran straight, which
makes each such pair a sample of the reference's surge dynamics
(`tests/golden/surge_golden.npz`, 425 samples, made by make_surge_golden.py).
All of them obey  u' = R u + (1 - R) F / X_u  with thrust F in [0, 2] N and
R = 0.678685866220174 = the stability polynomial of the Fehlberg 4(5) method's
5th-order solution at z = -X_u / m11 = -2 / 25.8, five steps of 1 s: this pins
the integrator, the step, the log interval, the linear surge damping over
inertia, and (324 samples saturate at exactly 0 N or exactly 2 N, the latter
with a steady speed of 1 m/s) the thrust limit.  Both oracles have to reproduce
the speed after the 5 steps to 1e-14."""
import os

import numpy as np

from gym_auv_amd import params
from oracle import auv_oracle as O
from oracle import c_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "surge_golden.npz")
TOL = 1e-14


def surge_ratio():
    z = -2.0 / 25.8
    r1 = 1 + z + z ** 2 / 2 + z ** 3 / 6 + z ** 4 / 24 + z ** 5 / 120 + z ** 6 / 2080   # RKF45, 5th-order solution
    return r1 ** 5


def samples():
    g = np.load(GOLDEN)
    u0, u1, psi = g["speed_before"], g["speed_after"], g["heading"]
    R = surge_ratio()
    thrust = 2.0 * (u1 - R * u0) / (1.0 - R)
    return u0, u1, psi, thrust


def test_reference_samples_follow_the_fehlberg_step():
    u0, u1, psi, thrust = samples()
    assert len(u0) == 425
    assert abs(surge_ratio() - 0.678685866220174) < 1e-15
    coasting = np.abs(thrust) < 1e-9
    full = np.abs(thrust - 2.0) < 1e-9
    assert coasting.sum() == 89 and full.sum() == 235
    assert thrust.min() > -1e-9 and thrust.max() < 2.0 + 1e-9          # the clipped surge command times 2 N
    # not Dormand-Prince, not RK4, not a finer step: their ratios differ from the data in the 9th digit
    ratio = (u1 / u0)[coasting]
    assert np.abs(ratio - surge_ratio()).max() < 1e-12
    z = -2.0 / 25.8
    others = {"exact": np.exp(5 * z), "dopri5": (1 + z + z**2/2 + z**3/6 + z**4/24 + z**5/120 + z**6/600) ** 5,
              "rk4": (1 + z + z**2/2 + z**3/6 + z**4/24) ** 5,
              "rkf45 h=0.5": (1 + (z/2) + (z/2)**2/2 + (z/2)**3/6 + (z/2)**4/24 + (z/2)**5/120 + (z/2)**6/2080) ** 10}
    for name, r in others.items():
        assert np.abs(ratio - r).min() > 5e-10, name


def _rollout_numpy(u0, psi, surge):
    n = len(u0)
    cfg = params.make_config(n, 0, 0)
    s = np.zeros((6, n))
    s[2], s[3] = psi, u0
    act = np.stack([surge, np.zeros(n)])
    for _ in range(5):
        s = O.step_dynamics(s, act, cfg)
    return s


def test_numpy_oracle_reproduces_the_reference_speeds():
    u0, u1, psi, thrust = samples()
    saturated = (np.abs(thrust) < 1e-9) | (np.abs(thrust - 2.0) < 1e-9)
    surge = np.where(np.abs(thrust - 2.0) < 1e-9, 1.0, np.where(np.abs(thrust) < 1e-9, 0.0, thrust / 2.0))
    s = _rollout_numpy(u0, psi, surge)
    speed = np.sqrt(s[3] ** 2 + s[4] ** 2)
    assert np.abs(speed - u1)[saturated].max() <= TOL       # the pin: nothing here is fitted to the data
    assert np.abs(speed - u1).max() <= TOL                   # partial thrust: the command is read off the sample
    assert np.array_equal(s[2], psi) and not s[4].any() and not s[5].any()   # straight: heading bit-identical


def test_c_oracle_reproduces_the_reference_speeds():
    u0, u1, psi, thrust = samples()
    n = len(u0)
    cfg = params.make_config(n, 0, 0)
    env = c_oracle.COracleEnv(cfg, params.ray_table(cfg["n_rays"]))
    env.set_scene(np.zeros((n, 0, 5)), np.zeros((n, 0, 6)), np.zeros((n, 0, 1, 2)), np.zeros((n, 0), dtype=np.int32), None)
    path = np.zeros((n, 6))
    path[:, 2], path[:, 3], path[:, 4], path[:, 5] = psi, np.cos(psi), np.sin(psi), 1e6
    env.set_path(path)
    init = np.zeros((6, n))
    init[2], init[3] = psi, u0
    env.reset(init)
    act = np.stack([np.clip(thrust / 2.0, 0.0, 1.0), np.zeros(n)])
    for _ in range(5):
        env.step(act)
    speed = np.sqrt(env.state[3] ** 2 + env.state[4] ** 2)
    assert np.abs(speed - u1).max() <= TOL
    env.close()
