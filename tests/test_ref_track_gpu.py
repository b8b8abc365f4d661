"""The device's position integration, obstacle distance and bearing against the
reference's own data, through the C ABI: 5,230 sightings of static obstacles in
consecutive rows of the reference's log (`tests/golden/track_golden.npz`; see
tests/test_ref_track_cpu.py for what they pin).  One env per sighting, 5 steps
with the logged command, on both sensing paths; dist and bearing afterwards are
the ones the reference logged."""
import numpy as np
import pytest

from gym_auv_amd import params
from tests.helpers import make_sim
from tests.test_ref_track_cpu import LOG_EVERY, check, scene_for, sightings

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1], ids=["fused", "allpairs"])
def test_device_tracks_the_reference_obstacles(mode):
    g = sightings()
    n = len(g["offset"])
    cfg = params.make_config(n, 1, 0)
    sim = make_sim(cfg, scene_for(g, cfg), lidar_mode=mode)
    sim.reset(g["state"].copy())
    act = np.stack([g["surge"], g["steer"]])
    for _ in range(LOG_EVERY):
        sim.step_host(act)
    check(g, sim.read("obst_dist"), sim.read("meta"), sim.read("collision"), sim.read("risk"), sim.read("own"))
    # and the scan sees the same obstacle: the ray nearest its bearing stops between the hull distance
    # and the tangent length
    ranges = sim.read("ranges")
    R = cfg["n_rays"]
    ray = np.rint((g["bearing_b"] + np.pi) * R / (2 * np.pi) - 1).astype(int) % R
    seen = g["dist_b"] + cfg["vessel_radius"] < cfg["sensor_range"] - 1.0
    got = ranges[np.arange(n), ray][seen]
    lo = (g["dist_b"] + cfg["vessel_radius"])[seen]
    assert (got >= lo - 1e-6).all() and (got <= lo + 1.0).all()
    sim.close()
