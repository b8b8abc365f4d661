"""The collision predicate and the ships' hull against the reference's own
near-contact rows (tests/golden/contact_golden.npz, from its metadata file).

The reference logs no contact flag, and none of its 36 logged episodes ends
near an obstacle; what its log does hold is 62 records of 14 close passes that
its vessel SURVIVED (the episodes went on for 23 to 467 more rows), 11 of them
with a logged `dist` below zero.  If the reference ends an episode on contact,
as a gym env does, these are rows on which a faithful predicate must not fire.
What the data support, and what is asserted here:
  (1) the predicate of this build (nearest hull distance less 3.255 m below
      contact_dist = -2, i.e. the hull within 1.255 m of the vessel's centre)
      fires on none of the rows, nor anywhere along the straight-line
      interpolation of the passes between rows;
  (2) contact at 3.255 m itself (contact_dist = 0, this build's earlier
      predicate) DOES fire on rows 7364-7366: the log contradicts it, and
      bounds the clearance from above at the smallest hull distance survived,
      2.53 m for this build's hull;
  (3) a hull reaching further ahead of the logged pose than about 1.8 sizes
      contains the vessel's centre at row 6986 (the vessel crossed the ship's
      bow at 1.88 sizes): the long outline that the shipped networks' error
      asks for (tests/test_ref_outline_cpu.py) is excluded by the log;
  (4) no logged episode ends with an obstacle nearer than 36 m: no episode of
      the log was ended by a contact, so the log holds no positive example.
Counting rows from 0 throughout.
"""
import numpy as np
import pytest

from gym_auv_amd import params
from oracle import auv_oracle as O
from tests import contact_scenes as CS


def sense(cfg, scene):
    env = O.OracleEnv(cfg, params.ray_table(cfg["n_rays"]))
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"], scene["poly_heading"])
    env.set_path(scene["path"])
    env.reset(scene["init_state"])
    return env


@pytest.fixture(scope="module")
def g():
    return CS.load()


def test_fixture_is_the_logs_close_passes(g):
    assert len(g["row"]) == 62 and int((g["record"][:, 1] < 0).sum()) == 11
    assert sorted(set(g["row"][g["record"][:, 1] < 0])) == [7362, 7363, 7364, 7365, 7366, 7696, 7697, 7869, 7870, 7871, 7872]
    assert g["static"].sum() == 7 and g["record"][g["static"], 1].min() > 1.1       # static passes: never below 1.11 m
    assert g["record"][:, 1].min() == pytest.approx(-7.536, abs=1e-3)
    assert g["rows_left"].min() == 23                    # every one of these passes was survived by 23 rows or more
    assert len(CS.passes(g)) == 14


def test_logged_distance_column_is_reproduced(g):
    """The scenes are the logged ones: the simulation's own `dist` column gives back the logged value."""
    cfg, scene = CS.build(g)
    env = sense(cfg, scene)
    slot = np.where(g["static"], 0, 1)
    assert np.abs(env.meta[np.arange(62), slot, 1] - g["record"][:, 1]).max() < 1e-12
    assert np.abs(O.wrap_angle(env.meta[np.arange(62), slot, 3] - g["record"][:, 3])).max() < 1e-12


def test_predicate_does_not_fire_on_rows_the_reference_survived(g):
    cfg, scene = CS.build(g)
    env = sense(cfg, scene)
    assert cfg["contact_dist"] == -2.0
    assert not env.collision.any() and not env.done.any()
    hull = env.min_dist + CS.CLEARANCE                    # distance from the vessel's centre to the hull
    assert hull.min() == pytest.approx(2.533, abs=1e-3) and g["row"][np.argmin(hull)] == 7365
    assert hull[g["static"]].min() == pytest.approx(1.110 + CS.CLEARANCE, abs=1e-3)
    # the earlier predicate, contact at the 3.255 m the logged distances are reduced by, fires on three of them
    cfg0, _ = CS.build(g, contact_dist=0.0)
    env0 = sense(cfg0, scene)
    assert sorted(g["row"][env0.collision != 0]) == [7364, 7365, 7366]
    # any clearance up to the smallest survived hull distance is consistent with the log, none above it
    for clearance, fires in ((2.5, False), (2.6, True)):
        cfg1, _ = CS.build(g, contact_dist=clearance - CS.CLEARANCE)
        assert bool(sense(cfg1, scene).collision.any()) == fires


def test_predicate_does_not_fire_between_the_logged_rows(g):
    """Rows are 5 steps apart.  Seen from the ship, the vessel's track through a pass is nearly straight between
    rows: nine interpolated places per gap, none in contact either."""
    ahead, abeam = CS.vessel_in_obstacle_frame(g)
    nearest = np.inf
    for f in np.linspace(0.1, 0.9, 9):
        pa, pb = ahead.copy(), abeam.copy()
        for run in CS.passes(g):
            for i, j in zip(run[:-1], run[1:]):
                pa[i] = (1 - f) * ahead[i] + f * ahead[j]
                pb[i] = (1 - f) * abeam[i] + f * abeam[j]
        cfg, scene = CS.build(g, places=(pa, pb))
        env = sense(cfg, scene)
        assert not env.collision.any(), f
        nearest = min(nearest, float((env.min_dist + CS.CLEARANCE)[~g["static"]].min()))
    print("smallest hull distance along the interpolated passes:", nearest)
    assert 2.0 < nearest < 2.54


def test_log_excludes_a_hull_reaching_far_ahead(g):
    """Row 6986: the vessel crossed ahead of a ship of size 8 at 1.88 sizes from its logged pose, 0.28 sizes off
    its axis, and the episode went on for 150 rows.  A hull from one size astern to four sizes ahead (where the
    shipped networks' error has its minimum for a hull scaled with the size) holds that point inside it."""
    ahead, abeam = CS.vessel_in_obstacle_frame(g)
    i = int(np.where(g["row"] == 6986)[0][0])
    assert g["size"][i] == 8 and g["rows_left"][i] == 150
    assert ahead[i] / 8 == pytest.approx(1.885, abs=5e-3) and abeam[i] / 8 == pytest.approx(0.277, abs=5e-3)

    def long_hull(w):
        w = np.asarray(w, dtype=np.float64)[..., None]
        x = np.concatenate([-w, 1.5 * w, 4 * w, 1.5 * w, -w], axis=-1)
        y = np.concatenate([-w / 2, -w / 2, 0 * w, w / 2, w / 2], axis=-1)
        return np.stack([x, y], axis=-1)

    cfg, scene = CS.build(g, hull=long_hull, contact_dist=-CS.CLEARANCE)     # contact = the centre inside the hull
    env = sense(cfg, scene)
    inside = g["row"][env.collision != 0]
    assert sorted(inside) == [6986, 6987]          # both rows of that crossing
    # this build's hull (2 sizes long, centred on the pose) holds the vessel's centre on no survived row
    cfg, scene = CS.build(g, contact_dist=-CS.CLEARANCE)
    assert not sense(cfg, scene).collision.any()


def test_no_logged_episode_ends_in_a_contact(g):
    assert len(g["ep_first"]) == 36 and int(g["ep_last"][-1]) == int(g["n_rows_file"]) - 1 == 10074
    assert g["ep_last_row_nearest"].min() > 36.0           # nearest obstacle at any episode's last row: 36.09 m
    assert (g["ep_nearest"] < 0).sum() == 2                # the two episodes with dist < 0 went on to end elsewhere
    lengths = g["ep_last"] - g["ep_first"] + 1
    assert lengths.min() == 26 and lengths.max() == 798    # no common length: not a fixed step limit either
