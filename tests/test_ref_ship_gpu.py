"""The device's moving-obstacle bookkeeping against the reference's own data,
through the C ABI: 134 ships of the reference's log (`tests/golden/ship_golden.npz`;
see tests/test_ref_ship_cpu.py), each followed over 15 steps; at every logged row
the metadata record and the risk the device holds are the logged ones, on both
sensing paths."""
import numpy as np
import pytest

from gym_auv_amd import params
from tests.helpers import make_sim
from tests.test_ref_ship_cpu import LOG_EVERY, ROWS, check_row, scene_for, ships

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1], ids=["fused", "allpairs"])
def test_device_reproduces_the_logged_ship_records(mode):
    g = ships()
    n = len(g["offset"])
    cfg = params.make_config(n, 0, 1, 5)
    sim = make_sim(cfg, scene_for(g, cfg), lidar_mode=mode)
    sim.reset(g["state"].copy())
    check_row(g, 0, sim.read("own"), sim.read("meta"), sim.read("risk"))
    act = np.stack([g["surge"], g["steer"]])
    for k in range(1, ROWS):
        for _ in range(LOG_EVERY):
            sim.step_host(act)
        check_row(g, k, sim.read("own"), sim.read("meta"), sim.read("risk"))
    sim.close()
