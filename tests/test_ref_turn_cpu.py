"""The sway-yaw block of the vessel dynamics against the reference's own data.

The reference's simulator is not in its checkout, but its dataset logs
[speed_OS, heading_OS] once per 5 simulator steps
(data/metadata_MovingObstaclesNoRules.csv; utils/explore_dataset.py:8).
`tests/golden/turn_golden.npz` (made by make_turn_golden.py) holds the 1,957
windows of 4 consecutive rows (a fifth of the log) through which the vessel
turned with the thruster at a limit and the rudder at zero (238 coasting, 1,584
at full thrust) or hard over (135): 7 logged numbers (4 speeds, 3 heading changes) that depend on 3
unknowns only, the body velocities (u, v, r) at the first row, which the fixture
carries as `start`.  Integrating 15 steps from that start has to give back all
7 numbers, to 1e-13 (the logs are met to 3e-15): this pins the inertia matrix of
the sway-yaw block, its linear damping and the velocity-dependent coupling
(m u r in sway, m x_g u r in yaw, none in surge) and, through the hard-over
windows, the yaw moment limit of 0.075 N m acting on yaw alone.  A textbook Coriolis matrix
with added mass in it misses the same numbers in the 3rd digit."""
import os

import numpy as np

from gym_auv_amd import params
from oracle import auv_oracle as O
from oracle import c_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "turn_golden.npz")
TOL = 1e-13
ROWS, LOG_EVERY = 4, 5


def windows():
    g = np.load(GOLDEN)
    return g["start"], g["speed"], g["heading"], np.stack([g["surge"], g["steer"]])


def compare(states, speed, heading):
    """states: list of [6,n] at rows 0..3 -> worst speed and heading deviation."""
    worst_s = worst_h = 0.0
    for k, s in enumerate(states):
        worst_s = max(worst_s, np.abs(np.sqrt(s[3] ** 2 + s[4] ** 2) - speed[k]).max())
        worst_h = max(worst_h, np.abs(O.wrap_angle(s[2] - heading[k])).max())
    return worst_s, worst_h


def test_fixture_is_over_determined_and_not_trivial():
    start, speed, heading, cmd = windows()
    n = 1957
    assert start.shape == (3, n) and speed.shape == (ROWS, n) and heading.shape == (ROWS, n)
    surge, steer = cmd
    assert ((surge == 0.0) & (steer == 0.0)).sum() == 238 and ((surge == 1.0) & (steer == 0.0)).sum() == 1584
    assert (steer == 1.0).sum() == 65 and (steer == -1.0).sum() == 70      # rudder hard over: the moment limit
    assert params.make_config(1, 0, 0)["moment_max"] == 0.075
    turn = np.abs(O.wrap_angle(heading[-1] - heading[0]))
    assert (turn > 1e-2).sum() >= 500 and turn.max() > 0.4         # real turns, up to 24 degrees per window
    assert np.abs(start[1]).max() > 0.15 and np.abs(start[2]).max() > 0.1   # sway 0.18 m/s, yaw rate 0.1 rad/s


def test_numpy_oracle_reproduces_the_reference_turns():
    start, speed, heading, cmd = windows()
    n = start.shape[1]
    cfg = params.make_config(n, 0, 0)
    s = np.zeros((6, n))
    s[2], s[3:] = heading[0], start
    act = cmd
    states = [s]
    for _ in range(ROWS - 1):
        for _ in range(LOG_EVERY):
            s = O.step_dynamics(s, act, cfg)
        states.append(s)
    ws, wh = compare(states, speed, heading)
    assert ws <= TOL and wh <= TOL


def test_textbook_coriolis_matrix_does_not_fit_the_reference_turns():
    """Guards the pin: the same windows reject M_A-based Coriolis terms."""
    start, speed, heading, cmd = windows()
    n = start.shape[1]
    cfg = params.make_config(n, 0, 0)
    mp = cfg["model"]
    m11, m22, m23 = 25.8, 33.8, 23.8 * 0.046

    def rhs(y):
        psi, u, v, r = y[2], y[3], y[4], y[5]
        c13 = -(m22 * v + m23 * r)
        c23 = m11 * u
        r1 = (2.0 * cmd[0] - c13 * r) - 2.0 * u
        r2 = -(c23 * r) - (7.0 * v + 0.1 * r)
        r3 = (0.075 * cmd[1] - (-(c13 * u) - c23 * v)) - (0.1 * v + 0.5 * r)
        return np.stack([np.cos(psi) * u - np.sin(psi) * v, np.sin(psi) * u + np.cos(psi) * v, r,
                         mp[O.MP_MINV11] * r1, mp[O.MP_MINV22] * r2 + mp[O.MP_MINV23] * r3,
                         mp[O.MP_MINV32] * r2 + mp[O.MP_MINV33] * r3])

    s = np.zeros((6, n))
    s[2], s[3:] = heading[0], start
    states = [s]
    for _ in range(ROWS - 1):
        for _ in range(LOG_EVERY):
            k = []
            for i in range(6):
                y = s if i == 0 else s + sum(O.RK_A[i][j] * k[j] for j in range(i))
                k.append(rhs(y))
            s = s + sum(O.RK_B5[j] * k[j] for j in range(6))
        states.append(s)
    ws, wh = compare(states, speed, heading)
    assert ws > 1e-4 and wh > 1e-3


def test_c_oracle_reproduces_the_reference_turns():
    start, speed, heading, cmd = windows()
    n = start.shape[1]
    cfg = params.make_config(n, 0, 0)
    env = c_oracle.COracleEnv(cfg, params.ray_table(cfg["n_rays"]))
    env.set_scene(np.zeros((n, 0, 5)), np.zeros((n, 0, 6)), np.zeros((n, 0, 1, 2)), np.zeros((n, 0), dtype=np.int32), None)
    path = np.zeros((n, 6))
    path[:, 2], path[:, 3], path[:, 4], path[:, 5] = heading[0], np.cos(heading[0]), np.sin(heading[0]), 1e6
    env.set_path(path)
    init = np.zeros((6, n))
    init[2], init[3:] = heading[0], start
    env.reset(init)
    act = cmd
    states = [init]
    for _ in range(ROWS - 1):
        for _ in range(LOG_EVERY):
            env.step(act)
        states.append(env.state.copy())
    ws, wh = compare(states, speed, heading)
    assert ws <= TOL and wh <= TOL
    env.close()
