"""The device's vessel dynamics against the reference's own data, through the
C ABI: 425 straight-line samples of the reference simulator's surge dynamics
(`tests/golden/surge_golden.npz`; see tests/test_ref_surge_cpu.py for what they
pin).  One env per sample, five steps with the sample's thrust command; the
speed afterwards has to be the one the reference logged, to 1e-14, on both
stepping paths."""
import numpy as np
import pytest

from gym_auv_amd import params
from tests.helpers import make_sim
from tests.test_ref_surge_cpu import TOL, samples

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1], ids=["fused", "allpairs"])
def test_device_reproduces_the_reference_speeds(mode):
    u0, u1, psi, thrust = samples()
    n = len(u0)
    cfg = params.make_config(n, 0, 0)
    path = np.zeros((n, 6))
    path[:, 2], path[:, 3], path[:, 4], path[:, 5] = psi, np.cos(psi), np.sin(psi), 1e6
    scene = {"circles": np.zeros((n, 0, 5)), "poly_pose": np.zeros((n, 0, 6)), "poly_verts": np.zeros((n, 0, 5, 2)),
             "poly_nv": np.zeros((n, 0), dtype=np.int32), "poly_heading": np.zeros((n, 0)), "path": path}
    sim = make_sim(cfg, scene, lidar_mode=mode)
    init = np.zeros((6, n))
    init[2], init[3] = psi, u0
    sim.reset(init)
    act = np.stack([np.clip(thrust / 2.0, 0.0, 1.0), np.zeros(n)])
    for _ in range(5):
        sim.step_host(act)
    s = sim.read("state")
    speed = np.sqrt(s[3] ** 2 + s[4] ** 2)
    saturated = (np.abs(thrust) < 1e-9) | (np.abs(thrust - 2.0) < 1e-9)
    assert saturated.sum() == 324
    assert np.abs(speed - u1)[saturated].max() <= TOL
    assert np.abs(speed - u1).max() <= TOL
    assert np.array_equal(s[2], psi)                     # straight: the heading is untouched
    own = sim.read("own")
    assert np.abs(own[:, 0] - u1).max() <= TOL           # and the logged speed_OS column is that speed
    sim.close()
