"""The LiDAR scan convention against the reference's own numbers (CPU oracle).

The reference's logged scans are a missing file, but tests/golden/scan_golden.npz
rebuilds the scenes of 9,711 of its 10,075 logged rows exactly (obstacle sizes
recovered to 1e-5 m, tests/golden/make_scan_golden.py).  Cast with this build's
LiDAR and fed to the reference's shipped networks (weights of
logs/trained_models/*.json), the scans must reproduce the labels the networks
were fitted to (row-wise max risk, utils/dataloader.py:113-114):
  (i)   on the reference's test split the error of every network lies within
        mean +- 1 sigma of the test errors the reference publishes
        (results_RL/_mse_*.txt);
  (ii)  on the rows the networks were trained on, this build's ray order (ray 0
        one step past dead astern, counter-clockwise) gives a lower error than
        all 23 other orders that start at a sector boundary (12 rotations by 15
        rays x 2 senses), for every network, each with a paired z-score >= 4;
        the same holds against a turn by 5 rays either way (z >= 3.5); and on
        the training rows that list static obstacles only -- where the ships'
        outline, which nothing in the reference fixes, plays no part -- against a
        turn by one or two sectors either way and against the clockwise order;
  (iii) a turn by 1 or 2 rays is NOT resolved (reported, not asserted): the
        convention is pinned to +-2 rays (+-4 degrees) and in its sense of rotation.
"""
import numpy as np
import pytest

from oracle import auv_oracle as O
from oracle import cnn_oracle as CN
from tests import scan_scenes as S

NETS = ("shallow", "deep2", "deep")


@pytest.fixture(scope="module")
def world():
    import os

    g = S.load("all")
    cfg, scene, label = S.build(g)
    rng, hit = O.lidar(scene["init_state"], scene["circles"], scene["poly_pose"], scene["poly_verts"],
                       scene["poly_nv"], S.ray_table(), cfg["sensor_range"])
    scans = np.asarray(O.closeness(rng, cfg["sensor_range"])).reshape(len(label), 180).astype(np.float32)
    gold = np.load(os.path.join(os.path.dirname(S.GOLDEN), "cnn_golden.npz"))
    W = {k: CN.golden_weights(gold, k) for k in NETS}
    return g, cfg, scene, label, scans, hit, W


def sq_err(scans, label, W):
    x = np.ascontiguousarray(scans).reshape(len(label), 1, 180)
    return np.stack([(CN.MODELS[k](x, W[k], "risk")[:, 0].astype(np.float64) - label) ** 2 for k in NETS], axis=1)


def test_fixture_covers_the_log(world):
    g = world[0]
    assert len(g["row"]) >= 9700 and int(g["n_rows_file"]) == 10075
    tr, va, te = S.split_masks(g)
    assert te.sum() >= 2980 and tr.sum() >= 4600
    assert g["size"][g["static"]].min() >= 15 and g["size"][g["static"]].max() <= 41
    assert g["size"][~g["static"]].min() >= 1 and g["size"][~g["static"]].max() <= 18
    # every listed obstacle is seen by some ray or hidden behind a nearer one: rows are never empty scans
    assert (world[5] >= 0).any(axis=1).mean() > 0.97


def test_reordered_scans_equal_recast_scans(world):
    g, cfg, scene, label, scans, hit, W = world
    sub = slice(0, 400)
    for shift, mirrored in ((15, False), (-2, False), (0, True), (45, True)):
        rng, _ = O.lidar(scene["init_state"][:, sub], scene["circles"][sub], scene["poly_pose"][sub],
                         scene["poly_verts"][sub], scene["poly_nv"][sub], S.ray_table(180, shift, mirrored),
                         cfg["sensor_range"])
        recast = np.asarray(O.closeness(rng, cfg["sensor_range"])).reshape(400, 180).astype(np.float32)
        same = recast == S.reorder(scans[sub], shift, mirrored)
        # the ray directions of a turned table differ from the original's in the last bits, so a grazing ray may flip
        assert same.mean() > 0.9995, (shift, mirrored, same.mean())


def test_error_on_the_test_split_is_the_published_one(world):
    g, cfg, scene, label, scans, hit, W = world
    _, _, te = S.split_masks(g)
    mse = sq_err(scans[te], label[te], W).mean(axis=0)
    lo = g["published_mse_mean"] - g["published_mse_std"]
    hi = g["published_mse_mean"] + g["published_mse_std"]
    print("test-split MSE (shallow, 2_deep, deep)", mse, "published", g["published_mse_mean"], "+-", g["published_mse_std"])
    assert (mse > lo).all() and (mse < hi).all(), (mse, lo, hi)
    # ... and over all rebuilt rows
    mse_all = sq_err(scans, label, W).mean(axis=0)
    assert (mse_all > lo).all() and (mse_all < hi).all(), mse_all


def test_ray_order_beats_every_other_sector_aligned_order(world):
    g, cfg, scene, label, scans, hit, W = world
    tr, _, _ = S.split_masks(g)
    x, y = scans[tr], label[tr]
    base = sq_err(x, y, W)
    n = len(y)
    worst = np.inf
    for mirrored in (False, True):
        for shift in list(range(0, 180, 15)) + [5, -5]:
            if shift == 0 and not mirrored:
                continue
            d = sq_err(S.reorder(x, shift, mirrored), y, W) - base     # > 0: the other order is worse
            z = d.mean(axis=0) / (d.std(axis=0) / np.sqrt(n))
            floor = 3.5 if abs(shift) == 5 and not mirrored else 4.0
            assert (z >= floor).all(), (shift, mirrored, z)
            worst = min(worst, z.min())
    print("smallest paired z-score of any other order against this build's:", worst)


def test_ray_order_on_rows_without_ships(world):
    g, cfg, scene, label, scans, hit, W = world
    tr, _, _ = S.split_masks(g)
    no_ships = np.array([g["static"][g["first"][r]:g["first"][r + 1]].all() for r in range(len(g["row"]))])
    m = tr & no_ships
    assert m.sum() >= 1400
    x, y = scans[m], label[m]
    base = sq_err(x, y, W)
    for shift, mirrored in ((15, False), (30, False), (165, False), (150, False), (0, True), (15, True), (165, True)):
        d = sq_err(S.reorder(x, shift, mirrored), y, W) - base
        z = d.mean(axis=0) / (d.std(axis=0) / np.sqrt(len(y)))
        assert (z >= 4.0).all(), (shift, mirrored, z)


def test_report_what_is_not_resolved(world):
    g, cfg, scene, label, scans, hit, W = world
    tr, _, _ = S.split_masks(g)
    x, y = scans[tr], label[tr]
    base = sq_err(x, y, W)
    for shift in (-2, -1, 1, 2):
        d = sq_err(S.reorder(x, shift), y, W) - base
        z = d.mean(axis=0) / (d.std(axis=0) / np.sqrt(len(y)))
        print(f"turn by {shift:+d} ray(s): paired z (shallow, 2_deep, deep) = {np.round(z, 1)}")
        assert (np.abs(z) < 8).all()   # far from the 13..31 of a wrong sector
    # the ship outline: the pose at mid-hull (this build) or at the centre of the stern section
    g2 = S.load("all")
    _, scene2, _ = S.build(g2, outline="stern")
    rng, _ = O.lidar(scene2["init_state"], scene2["circles"], scene2["poly_pose"], scene2["poly_verts"],
                     scene2["poly_nv"], S.ray_table(), cfg["sensor_range"])
    alt = np.asarray(O.closeness(rng, cfg["sensor_range"])).reshape(len(label), 180).astype(np.float32)
    print("all-row MSE, hull centred on the pose:", sq_err(scans, label, W).mean(axis=0),
          "; pose at the stern section:", sq_err(alt, label, W).mean(axis=0))
