"""The device's collision flag on the reference's near-contact rows, through
the C ABI, on both sensing paths (k_sense; k_encounter of the all-pairs path):
tests/test_ref_contact_cpu.py says what the rows are and what they support."""
import numpy as np
import pytest

from tests import contact_scenes as CS
from tests.helpers import assert_all_equal, make_oracle, make_sim

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [0, 1], ids=["fused", "allpairs"])
def test_device_flags_no_contact_on_rows_the_reference_survived(mode):
    g = CS.load()
    cfg, scene = CS.build(g)
    sim = make_sim(cfg, scene, lidar_mode=mode)
    env = make_oracle(cfg, scene)
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    assert_all_equal(sim, env, "near-contact rows")
    assert not sim.read("collision").any() and not sim.read("done").any()
    slot = np.where(g["static"], 0, 1)
    assert np.abs(sim.read("meta")[np.arange(62), slot, 1] - g["record"][:, 1]).max() < 1e-12
    hull = sim.read("min_dist") + CS.CLEARANCE
    assert hull.min() == pytest.approx(2.533, abs=1e-3)
    sim.close()
    # contact at the 3.255 m of the logged distances fires on three survived rows: the log contradicts it
    cfg0, _ = CS.build(g, contact_dist=0.0)
    sim0 = make_sim(cfg0, scene, lidar_mode=mode)
    env0 = make_oracle(cfg0, scene)
    sim0.reset(scene["init_state"])
    env0.reset(scene["init_state"])
    assert_all_equal(sim0, env0, "near-contact rows, contact at the logged clearance")
    assert sorted(g["row"][sim0.read("collision") != 0]) == [7364, 7365, 7366]
    sim0.close()


def test_contact_threshold_is_the_configured_one():
    """The flag is `min_dist < contact_dist` on every path: thresholds either side of an env's distance."""
    g = CS.load()
    cfg, scene = CS.build(g)
    sim = make_sim(cfg, scene)
    sim.reset(scene["init_state"])
    d = np.sort(sim.read("min_dist"))
    sim.close()
    for k in (0, 10, 40):
        thr = 0.5 * (d[k] + d[k + 1])
        cfg_k, _ = CS.build(g, contact_dist=float(thr))
        for mode in (0, 1):
            s = make_sim(cfg_k, scene, lidar_mode=mode)
            s.reset(scene["init_state"])
            assert int(s.read("collision").sum()) == k + 1
            s.close()
