"""What the reference's data say about the ships' outline, and why this build
keeps its own (a hull two sizes long, one wide, centred on the logged pose).

The outline is not logged.  Two kinds of evidence exist:
  * the shipped networks (logs/trained_models/*.json), trained on the
    reference's real scans: fed the rebuilt rows (tests/golden/scan_golden.npz)
    cast with a candidate outline, their error against the labels is a
    criterion.  Over hulls scaled with the ship's size (beam, length, offset
    ahead; scratch/outline_fit3.py: 144 grid points) it has its minimum at
    about beam 1, length 5-6, mid-hull 1.5-2 sizes ahead for the two small
    networks -- a hull from one size astern to four or five ahead -- and keeps
    falling to the grid's edge for the deep one.
  * the near-contact rows (tests/test_ref_contact_cpu.py): the vessel crossed
    a ship's bow 1.88 sizes ahead of its pose and sailed on.
The two disagree, and the first is not to be trusted on its own: an extension
of the bow by 80 s of the ship's own motion -- no hull at all, a streak that
tells the network where the ship is going -- lowers the error further than any
hull scaled with the size.  The criterion rewards whatever correlates with the
label, which depends on the relative motion a single scan does not show; it
cannot identify an outline.  Asserted here on a handful of grid points."""
import os

import numpy as np
import pytest

from oracle import auv_oracle as O
from oracle import cnn_oracle as CN
from tests import scan_scenes as S

NETS = ("shallow", "deep2", "deep")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def world():
    g = S.load("all")
    c = np.load(os.path.join(ROOT, "tests", "golden", "cri_golden.npz"))
    tr, _, _ = S.split_masks(g)
    has_ship = np.array([(~g["static"][g["first"][r]:g["first"][r + 1]]).any() for r in range(len(g["row"]))])
    g["keep"] = tr & has_ship
    cfg, scene, label = S.build(g)
    first_rec = np.searchsorted(c["row"], np.arange(int(g["n_rows_file"]) + 1))
    size, speed = np.zeros(scene["poly_nv"].shape), np.zeros(scene["poly_nv"].shape)
    for e, r in enumerate(np.where(g["keep"])[0]):
        k = 0
        for j, o in enumerate(range(int(g["first"][r]), int(g["first"][r + 1]))):
            if g["static"][o]:
                continue
            # the ship's own speed from its record: velocity = own (mirrored when the headings differ by more than
            # pi, as the reference's columns have it) + rel_speed (cos course, sin course), taken along its heading
            q = first_rec[int(g["row"][r])] + j
            psi, hts = c["own_heading"][q], c["heading_ts"][q]
            s = c["own_speed"][q] * np.array([np.cos(psi), np.sin(psi)])
            h = np.array([np.cos(hts), np.sin(hts)])
            if hts - psi > np.pi or hts - psi <= -np.pi:
                s = 2 * (s @ h) * h - s
            v = s + c["rel_speed"][q] * np.array([np.cos(c["course"][q]), np.sin(c["course"][q])])
            size[e, k], speed[e, k] = g["size"][o], v @ h
            k += 1
    gold = np.load(os.path.join(os.path.dirname(S.GOLDEN), "cnn_golden.npz"))
    return cfg, scene, label, size, speed, {k: CN.golden_weights(gold, k) for k in NETS}


def mse(world, stern, bow, beam=1.0):
    """Error of the three networks with every ship drawn from `stern` to `bow` (metres ahead of its pose, arrays
    like the sizes), `beam` sizes wide, the bow tapering over the front half."""
    cfg, scene, label, size, speed, W = world
    w = size[..., None]
    st, bw = stern[..., None], bow[..., None]
    mid = 0.5 * (st + bw)
    x = np.concatenate([st, mid, bw, mid, st], axis=-1)
    y = np.concatenate([-beam * w / 2, -beam * w / 2, 0 * w, beam * w / 2, beam * w / 2], axis=-1)
    verts = np.stack([x, y], axis=-1) * (scene["poly_nv"] > 0)[..., None, None]
    rng, _ = O.lidar(scene["init_state"], scene["circles"], scene["poly_pose"], verts, scene["poly_nv"], S.ray_table(),
                     cfg["sensor_range"])
    xx = np.asarray(O.closeness(rng, cfg["sensor_range"])).reshape(len(label), 1, 180).astype(np.float32)
    return np.array([np.mean((CN.MODELS[k](xx, W[k], "risk")[:, 0].astype(np.float64) - label) ** 2) for k in NETS])


def test_network_error_cannot_identify_the_outline(world):
    cfg, scene, label, size, speed, W = world
    assert len(label) > 3000 and 0.05 < speed[size > 0].min() and speed[size > 0].max() < 1.3
    own = mse(world, -1.0 * size, 1.0 * size)            # this build's hull: 2 x 1 sizes, centred
    stern_kind = mse(world, -0.5 * size, 1.5 * size)     # the same hull with the pose in its stern section
    long_hull = mse(world, -1.0 * size, 4.0 * size)      # where the size-scaled family has its minimum
    longer = mse(world, -1.0 * size, 7.0 * size)
    streak = mse(world, -1.0 * size, 1.0 * size + 80.0 * speed)   # this build's hull + 80 s of the ship's motion
    print("MSE (shallow, 2_deep, deep): own hull", own, "stern kind", stern_kind, "long hull", long_hull,
          "longer", longer, "own hull + 80 s streak", streak)
    assert (stern_kind < own).all() and (long_hull < stern_kind).all()       # the criterion asks for length ahead ...
    assert (longer[:2] > long_hull[:2]).all()                                # ... with a minimum for the small networks
    assert (streak < long_hull).all()                                         # ... but prefers what no hull is
    assert (streak < 0.9 * own).all()
