"""HIP forward passes of the LiDAR CNNs against the reference-generated golden
outputs and the numpy oracle, and the zero-copy hand-off of a live scan."""
import os

import numpy as np
import pytest

from gym_auv_amd import _lib, cnn, scenarios
from oracle import cnn_oracle as N
from tests.helpers import make_sim

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "cnn_golden.npz")
TOL = 5e-6   # fp32; summation order differs from torch's CPU kernels


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("tag", ["shallow", "deep2", "deep"])
@pytest.mark.parametrize("head", ["risk", "features"])
def test_hip_cnn_matches_reference_golden(gold, tag, head):
    import torch

    net = cnn.LidarCNN(tag, N.golden_weights(gold, tag))
    x = torch.from_numpy(gold["x"]).cuda()
    y = net(x, head).cpu().numpy()
    ref = gold[f"{tag}_{head}"]
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= TOL, np.abs(y - ref).max()
    net.close()


@pytest.mark.parametrize("tag", ["shallow", "deep2", "deep"])
def test_hip_cnn_ragged_batches_against_oracle(gold, tag):
    import torch

    W = N.golden_weights(gold, tag)
    net = cnn.LidarCNN(tag, W)
    rng = np.random.default_rng(3)
    for B in (1, 15, 16, 17, 1000, 4099):
        x = rng.uniform(0, 1, (B, 1, 180)).astype(np.float32)
        for head in ("risk", "features"):
            y = net(torch.from_numpy(x).cuda(), head).cpu().numpy()
            assert np.abs(y - N.MODELS[tag](x, W, head)).max() <= TOL
    assert net(torch.empty((0, 1, 180), dtype=torch.float32, device="cuda")).shape == (0, 1)
    net.close()


def test_scan_feeds_cnn_zero_copy(gold):
    """cfg3's hand-off: the perception buffer written by k_sense is read in
    place by the shallow CNN; features + navigation = the extractor output of
    radarCNN.py:270-280."""
    import torch

    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=2048)
    sim = make_sim(cfg, scene)
    st = torch.cuda.current_stream()
    sim.reset(scene["init_state"], stream=st)
    obs = {"perception": sim.tensor("perception"), "navigation": sim.tensor("navigation")}
    W = N.golden_weights(gold, "shallow")
    ext = cnn.PerceptionNavigationExtractor("shallow", W)
    assert ext.features_dim == 18
    acts = torch.from_numpy(scenarios.random_actions(2048, 3, seed=8)).cuda()
    for k in range(3):
        sim.step_device(acts[k].data_ptr(), stream=st)
        feats = ext(obs)
        risk = ext.cnn(obs["perception"], "risk")
        st.synchronize()
        scans = sim.read("perception")
        assert obs["perception"].data_ptr() == sim.buffer("perception")[0]
        assert np.abs(feats[:, :12].cpu().numpy() - N.shallow(scans, W, "features")).max() <= TOL
        assert np.array_equal(feats[:, 12:].cpu().numpy(), sim.read("navigation")[:, 0, :])
        assert np.abs(risk.cpu().numpy() - N.shallow(scans, W, "risk")).max() <= TOL
    sim.close()


@pytest.mark.parametrize("mode,tile", [(_lib.LIDAR_RASTER, 0), (_lib.LIDAR_RASTER, 4), (_lib.LIDAR_RASTER, 2),
                                       (_lib.LIDAR_RASTER, 12), (_lib.LIDAR_ALL_PAIRS, 0)],
                         ids=["raster", "raster-tile4", "raster-tile2-general-kernel", "raster-tile12-queued-kernel",
                              "allpairs"])
def test_attached_cnn_equals_separate_kernel(gold, mode, tile):
    """A network attached to the batch is evaluated inside the sensing pass
    (in-kernel on the rasterising path); its output must equal the separate
    forward pass on the perception buffer bit for bit, for every network and
    head, on a batch that does not fill its last tile, after reset and step."""
    import torch

    V = 1001
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=V)
    sim = make_sim(cfg, scene, lidar_mode=mode)
    if tile:
        # 4 envs per tile: the kernel's compile-time shape; 2: the general kernel; 12 (192 obstacles, three rounds
        # of 64): the general kernel with the work queue, where the network's input takes the place of the
        # obstacle records only after the per-env summary has read them
        sim.set_option(_lib.OPT_RASTER_TILE, tile)
    st = torch.cuda.current_stream()
    acts = torch.from_numpy(scenarios.random_actions(V, 2, seed=5)).cuda()
    for tag in ("shallow", "deep2", "deep"):
        W = N.golden_weights(gold, tag)
        net = cnn.LidarCNN(tag, W)
        for head in ("features", "risk"):
            sim.attach_cnn(net, head)
            assert sim.buffer("cnn_out")[1] == (V, net.out_dim(head))
            sim.reset(scene["init_state"], stream=st)
            for k in range(3):
                if k:
                    sim.step_device(acts[k - 1].data_ptr(), stream=st)
                want = net(sim.tensor("perception"), head)
                st.synchronize()
                got = sim.read("cnn_out")
                assert np.array_equal(got.view(np.uint32), want.cpu().numpy().view(np.uint32)), (tag, head, k)
                assert np.abs(got - N.MODELS[tag](sim.read("perception"), W, head)).max() <= TOL
        net.close()   # the batch keeps its own copy of the weights
    sim.step_device(acts[1].data_ptr(), stream=st)
    st.synchronize()
    sim.attach_cnn(None)
    sim.close()


def test_attached_cnn_rejects_other_scan_widths(gold):
    cfg, scene = scenarios.make_workload("cfg5_dense_720", n_envs=8)
    sim = make_sim(cfg, scene)
    net = cnn.LidarCNN("shallow", N.golden_weights(gold, "shallow"))
    with pytest.raises(Exception, match="180"):
        sim.attach_cnn(net, "features")
    net.close()
    sim.close()


def test_dataset_logger_writes_the_oracles_rows(tmp_path):
    """config 5's use: regenerate the three dataset files from a running batch.
    The logger (page-locked reads every 5th step, native formatter) must write
    byte for byte what the same writer produces from the oracle's buffers, and
    the loader must read it back (utils/dataloader.py:55-56,104-120 semantics)."""
    from gym_auv_amd import dataset as D
    from gym_auv_amd import params
    from tests.helpers import make_oracle

    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=150)
    sim = make_sim(cfg, scene)
    env = make_oracle(cfg, scene)
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    acts = scenarios.random_actions(150, 15, seed=4)
    dev_dir, ora_dir = str(tmp_path / "dev"), str(tmp_path / "ora")
    wd, wo = D.DatasetWriter(dev_dir), D.DatasetWriter(ora_dir)
    logger = D.DatasetLogger(sim, wd, every=5)
    rows = 0
    for k in range(15):
        sim.step_host(acts[k])
        env.step(acts[k])
        rows += logger.after_step()
        if (k + 1) % 5 == 0:
            # no step of this rollout is a multiple of 25: the lists are those of the first logged step
            if k + 1 == 5:
                tracked = np.isfinite(env.obst_dist) & (env.meta[:, :, 1] < cfg["sensor_range"])
            tracked = tracked & np.isfinite(env.obst_dist)
            wo.write(env.ranges, env.own, env.meta, env.risk, tracked)
    wd.close()
    wo.close()
    assert rows == wo.rows > 0
    for kind in ("LiDAR", "risk", "metadata"):
        assert open(wd.paths[kind], "rb").read() == open(wo.paths[kind], "rb").read(), kind
    data = D.load_lidar_dataset(wd.paths["LiDAR"], wd.paths["risk"], mode="max")
    n_train, n_val, n_test = D.split_sizes(rows)
    assert data["train"][0].shape == (n_train, 1, 180) and data["test"][1].shape == (n_test, 1)
    # a page-locked read returns the same bytes as a plain one
    assert np.array_equal(sim.read("ranges", pinned=True), sim.read("ranges"))
    # the files have the relation the reference's own files have: the reference's risk index
    # applied to a metadata row (read back from the text) gives the risk row
    from oracle import ref_cri

    meta_rows, risk_rows = D.read_ragged(wd.paths["metadata"]), D.read_ragged(wd.paths["risk"])
    assert len(meta_rows) == len(risk_rows) == rows
    n_static = n_moving = 0
    for m, r in zip(meta_rows, risk_rows):
        speed_os, heading_os, rec = D.parse_metadata_row(m)
        assert len(rec) == len(r) >= 1
        got = ref_cri.cri_rows(np.full(len(rec), speed_os), np.full(len(rec), heading_os), rec[:, 0], rec[:, 1],
                               rec[:, 2], rec[:, 3], rec[:, 4])
        assert np.abs(got - r).max() < 1e-9
        st = np.array([D.is_static(x, speed_os) for x in rec])
        n_static += int(st.sum())
        n_moving += int((~st).sum())
    assert n_static > 100 and n_moving > 100
    sim.close()


def test_regenerated_dataset_resembles_the_reference_file(tmp_path):
    """Scenes from the generator with the sizes and speeds of the reference's logged rows, listed by the
    reference's rule, logged every 5 steps: the bucket table of utils/explore_dataset.py:40-53 over the regenerated
    metadata / risk files must resemble the reference file's (tests/golden/scan_golden.npz holds its counts).
    Tolerances are absolute differences of bucket fractions; the vessel here is steered at random, the
    reference's by a trained agent, so only the coarse shape is comparable."""
    import torch

    from gym_auv_amd import dataset as D
    from gym_auv_amd import params
    from gym_auv_amd.sim import AuvSim

    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "scan_golden.npz"))
    V = 600
    cfg = params.make_config(V, 8, 8, 5, max_steps=400)
    sim = AuvSim(cfg)
    sim.generate_scene(2026, scenarios.REFERENCE_LIKE)
    sim.set_option(_lib.OPT_AUTO_RESET, 1)
    sim.set_option(_lib.OPT_REGENERATE, 1)
    sim.reset(None)
    w = D.DatasetWriter(str(tmp_path))
    logger = D.DatasetLogger(sim, w)
    rng = np.random.default_rng(6)
    act = np.zeros((2, V))
    for k in range(150):
        if k % 10 == 0:   # a helmsman who holds a command for a while: mostly ahead at 60-100 % thrust
            act[0], act[1] = rng.uniform(0.6, 1.0, V), rng.uniform(-0.3, 0.3, V)
        sim.step_host(act)
        logger.after_step()
    w.close()
    meta, risk = D.read_ragged(w.paths["metadata"]), D.read_ragged(w.paths["risk"])
    assert len(meta) == len(risk) == w.rows > 5000
    table = {k: np.zeros_like(ref[k]) for k in ref.files if k.startswith("bucket_")}
    far = 0.0
    for m, r in zip(meta, risk):
        speed_os, _, rec = D.parse_metadata_row(m)
        table["bucket_count"][min(len(rec), 8)] += 1
        far = max(far, rec[:, 1].max())
        for x, rk in zip(rec, r):
            name = "static" if D.is_static(x, speed_os) else "dynamic"
            table[f"bucket_{name}_dist"][np.searchsorted([50.0, 100.0], x[1], side="right")] += 1
            table[f"bucket_{name}_risk"][min(int(rk / 0.2), 4)] += 1
            table[f"bucket_{name}_speed"][min(int(x[2] / 0.5), 3)] += 1
    assert far > 150.0 + 1.0      # lists have memory: obstacles stay listed beyond the sensor range, as in the reference
    frac = lambda a: a / a.sum()
    n_static = table["bucket_static_dist"].sum()
    n_dynamic = table["bucket_dynamic_dist"].sum()
    ref_static = ref["bucket_static_dist"].sum() / (ref["bucket_static_dist"].sum() + ref["bucket_dynamic_dist"].sum())
    assert abs(n_static / (n_static + n_dynamic) - ref_static) < 0.15
    report = {}
    for k, tol in (("bucket_static_dist", 0.15), ("bucket_dynamic_dist", 0.15), ("bucket_static_risk", 0.15),
                   ("bucket_dynamic_risk", 0.2), ("bucket_static_speed", 0.3), ("bucket_count", 0.15)):
        d = np.abs(frac(table[k]) - frac(ref[k])).max()
        report[k] = (np.round(frac(table[k]), 3).tolist(), np.round(frac(ref[k]), 3).tolist())
        assert d < tol, (k, report[k])
    print("regenerated vs reference bucket fractions:", report)
    sim.close()


def test_vec_env_carries_the_attached_features(gold):
    import torch

    from gym_auv_amd.env import VecAuvEnv

    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=257)
    W = N.golden_weights(gold, "shallow")
    net = cnn.LidarCNN("shallow", W)
    env = VecAuvEnv(cfg, scene, auto_reset="device", cnn=net)
    obs = env.reset()
    ext = cnn.PerceptionNavigationExtractor("shallow", W)
    for k in range(3):
        want = ext({"perception": obs["perception"], "navigation": obs["navigation"]})
        got = torch.cat([obs["cnn"], obs["navigation"][:, 0, :]], dim=1)
        assert torch.equal(got, want)
        a = torch.rand(257, 2, device="cuda")
        obs, rew, done, info = env.step(a)
    env.close()
