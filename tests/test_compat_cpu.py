"""The reference's training script against this package, with only its import
lines changed (trainer.py:6-7,11 -> gym_auv_amd.compat): the same calls with
the same arguments as its `__main__` (trainer.py:113-168), on a dataset small
enough for the CPU suite: the first 300 rows of the reference's risk file and
a LiDAR file cast here from the scenes of those rows
(tests/golden/scan_golden.npz).  The plots of that script are left out."""
import os
import sys

import numpy as np
import pytest
import torch

from gym_auv_amd import compat, dataset
from oracle import auv_oracle as O
from tests import scan_scenes as S

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    """LiDAR_/risk_ files in the reference's text format: the rebuilt rows among the first 300 of its dataset."""
    g = S.load("all")
    g["keep"] = g["row"] < 300
    rows = g["row"][g["keep"]]
    assert len(rows) > 250
    cfg, scene, label = S.build(g)
    env = O.OracleEnv(cfg, S.ray_table())
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"], scene["poly_heading"])
    env.set_path(scene["path"])
    env.reset(scene["init_state"])
    d = tmp_path_factory.mktemp("ref_files")
    path_x, path_y = str(d / "LiDAR_MovingObstaclesNoRules.csv"), str(d / "risk_MovingObstaclesNoRules.csv")
    np.savetxt(path_x, np.asarray(env.ranges).reshape(len(rows), 180))
    risk_lines = open(os.path.join(G, "risk_rows.csv")).read().splitlines(keepends=True)
    with open(path_y, "w") as f:
        f.writelines(risk_lines[r] for r in rows)
    return path_x, path_y, label


def test_the_reference_training_script_runs_with_its_imports_changed(files, capsys):
    from gym_auv_amd.compat import LidarCNN_deep, LidarCNN_2_deep      # was: from models.lidar_cnn_deep import ...
    from gym_auv_amd.compat import LidarCNN_shallow                    # was: from models.lidar_cnn_shallow import ...
    from gym_auv_amd.compat import load_LiDARDataset                   # was: from utils.dataloader import ...
    from gym_auv_amd.compat import Trainer                             # the class trainer.py:18 defines

    path_x, path_y, label = files
    torch.manual_seed(2)
    data_train, data_val, data_test, dataloader_train, dataloader_val, dataloader_test = load_LiDARDataset(
        path_x, path_y, mode='max', batch_size=16, train_test_split=0.7, train_val_split=0.3, shuffle=True)
    cnn_shallow = LidarCNN_shallow(n_sensors=180, output_channels=[1], kernel_size=45)
    cnn_deep = LidarCNN_deep(n_sensors=180, output_channels=[2, 4, 4, 6], kernel_size=9)
    cnn_not_so_deep = LidarCNN_2_deep(n_sensors=180, output_channels=[3, 2, 1], kernel_size=45)
    trainer = Trainer(model=cnn_deep, epochs=1, learning_rate=0.0005, dataloader_train=dataloader_train,
                      dataloader_val=dataloader_val, optimizer='adam')       # 14 epochs in the script
    trainer.train()
    assert "EPOCH 1" in capsys.readouterr().out
    assert len(trainer.training_loss) == len(trainer.validation_loss) == 1
    assert np.isfinite(trainer.training_loss + trainer.validation_loss).all()
    trainer.model.eval()
    with torch.no_grad():
        y_pred = trainer.model(data_test.X)
    assert y_pred.shape == data_test.y.shape == (len(data_test), 1)
    mse = float(((y_pred - data_test.y) ** 2).mean())
    assert np.isfinite(mse)
    # the split and the tensors are the reference's: sizes of utils/dataloader.py:63-67, closeness in [0,1],
    # labels = the rows' largest risk
    n = len(label)
    tr = int(0.7 * n)
    va = int(0.3 * tr)
    assert (len(data_train), len(data_val), len(data_test)) == (tr - va, va, n - tr)
    assert data_train.X.shape[1:] == (1, 180) and data_train.X.dtype == data_train.y.dtype == torch.float32
    assert 0.0 <= float(data_train.X.min()) and float(data_train.X.max()) <= 1.0
    assert np.array_equal(torch.cat([data_train.y, data_val.y, data_test.y])[:, 0].numpy(), label.astype(np.float32))
    assert dataloader_train.batch_size == dataloader_val.batch_size == 16 and dataloader_test.batch_size == 1
    assert dataloader_train.drop_last and dataloader_val.drop_last
    # the evaluate branch (trainer.py:180-193): the reference's weight files load into these classes
    gold = np.load(os.path.join(G, "cnn_golden.npz"))
    for net, tag in ((cnn_shallow, "shallow"), (cnn_not_so_deep, "deep2"), (cnn_deep, "deep")):
        pre = f"w_{tag}/"
        net.load_state_dict({k[len(pre):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(pre)})
        net.eval()
        with torch.no_grad():
            out = net(torch.from_numpy(gold["x"])).numpy()
        assert np.abs(out - gold[f"{tag}_risk"]).max() <= 1e-6


def test_constructor_defaults_are_the_references():
    """trainer.py:25,35-38: SGD with momentum 0.9 unless 'adam' is asked for, which is AdamW."""
    X = np.zeros((32, 180))
    d = compat.LiDARDataset(X, np.zeros(32))
    ld = torch.utils.data.DataLoader(d, batch_size=16, shuffle=True, drop_last=True)
    t = compat.Trainer(compat.LidarCNN_shallow(), 1, 0.01, ld, ld, device="cpu")
    assert isinstance(t.optimizer, torch.optim.SGD) and t.optimizer.defaults["momentum"] == 0.9
    t = compat.Trainer(compat.LidarCNN_shallow(), 1, 0.01, ld, ld, optimizer="adam", device="cpu")
    assert isinstance(t.optimizer, torch.optim.AdamW)
    # prev_steps stacks earlier scans behind each scan (utils/dataloader.py:30-33)
    assert compat.LiDARDataset(X, np.zeros(32), prev_steps=2).X.shape == (32, 1, 180, 3)


def test_epoch_order_is_the_order_the_loader_yields():
    """`epoch_order` must draw from torch's generator exactly as iterating the loader does: that is what lets the
    GPU path train on the batches the reference's loop would have seen."""
    X = np.arange(70, dtype=np.float64)[:, None] * np.ones((1, 180))
    d = compat.LiDARDataset(X, np.arange(70, dtype=np.float64))
    ld = torch.utils.data.DataLoader(d, batch_size=16, shuffle=True, drop_last=True)
    torch.manual_seed(2)
    seen = [torch.cat([y[:, 0] for _, y in ld]).long() for _ in range(3)]          # three epochs
    torch.manual_seed(2)
    drawn = [compat.epoch_order(ld) for _ in range(3)]
    for a, b in zip(seen, drawn):
        assert len(a) == 64 and torch.equal(a, b)
    assert not torch.equal(drawn[0], drawn[1])


def test_reference_import_lines_work_after_install(files):
    """With `install_reference_names()` even the import lines of trainer.py:6-7,11 stay as they are."""
    if any(m in sys.modules for m in ("models", "utils")):
        pytest.skip("another 'models' / 'utils' package is already imported in this process")
    compat.install_reference_names()
    try:
        from models.lidar_cnn_deep import LidarCNN_deep, LidarCNN_2_deep
        from models.lidar_cnn_shallow import LidarCNN_shallow
        from utils.dataloader import load_LiDARDataset

        assert LidarCNN_deep is compat.LidarCNN_deep and LidarCNN_2_deep is compat.LidarCNN_2_deep
        assert LidarCNN_shallow is compat.LidarCNN_shallow and load_LiDARDataset is compat.load_LiDARDataset
        out = load_LiDARDataset(files[0], files[1], 16, mode="sum")
        assert len(out) == 6
    finally:
        for m in [m for m in sys.modules if m == "models" or m.startswith("models.") or m == "utils" or m.startswith("utils.")]:
            if getattr(sys.modules[m], "__gym_auv_amd_compat__", False):
                del sys.modules[m]


# ---- the RL side, radarCNN.py:10-280 ----
def _golden_state(tag):
    gold = np.load(os.path.join(G, "cnn_golden.npz"))
    pre = f"w_{tag}/"
    return gold, {k[len(pre):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(pre)}


@pytest.mark.parametrize("tag,cls,kw", [
    ("shallow", "LidarCNN_shallow_pretrained", dict(n_sensors=180, output_channels=[1], kernel_size=45, padding=15,
                                                    stride=15, features_dim=12)),
    ("deep2", "LidarCNN_not_so_deep_pretrained", dict(n_sensors=180, output_channels=[3, 2, 1], kernel_size=45,
                                                      features_dim=12)),
    ("deep", "LidarCNN_deep_pretrained", dict(n_sensors=180, output_channels=[2, 4, 4, 6], kernel_size=9,
                                              features_dim=8))])
def test_pretrained_extractors_take_the_shipped_weights_and_give_the_references_features(tag, cls, kw):
    """Constructed with the arguments radarCNN.py:219-239 passes, loaded the way radarCNN.py:243-250 loads (the
    keys the shipped file shares with the extractor, strictly), they return what the reference's own modules
    returned for the fixture's scans (tests/golden/make_cnn_golden.py)."""
    from gym_auv_amd.env import Box

    gold, shipped = _golden_state(tag)
    net = getattr(compat, cls)(observation_space=Box(0.0, 1.0, (1, 180)), **kw)
    have = net.state_dict()
    pretrained = {k: v for k, v in shipped.items() if k in have}
    assert set(pretrained) == set(have) and len(pretrained) < len(shipped)   # the risk head's layers stay behind
    net.load_state_dict(pretrained)
    assert net.features_dim == kw["features_dim"]
    with torch.no_grad():
        got = net(torch.from_numpy(gold["x"])).numpy()
    assert got.shape == gold[f"{tag}_features"].shape
    assert np.abs(got - gold[f"{tag}_features"]).max() <= 1e-6


def test_perception_navigation_extractor_as_the_policy_builds_it(tmp_path, capsys):
    """radarCNN.py:202-280 with the env's Dict observation space: 12 frozen
    CNN features + the 6 navigation features."""
    import copy

    from gym_auv_amd.env import Box, DictSpace

    gold, shipped = _golden_state("shallow")
    path = str(tmp_path / "model_shallow_pretrained.json")
    torch.save(shipped, path)                     # a torch archive under a .json name, as the reference ships it
    space = DictSpace({"perception": Box(0.0, 1.0, (1, 180)), "navigation": Box(-np.inf, np.inf, (1, 6))})
    ext = compat.PerceptionNavigationExtractor(space, sensor_dim=180, features_dim=12, cnn_path=path)
    assert "Pretraind layers: ['feature_extractor.0.weight', 'feature_extractor.0.bias']" in capsys.readouterr().out
    assert ext.features_dim == 18 and not any(p.requires_grad for p in ext.parameters())
    nav = torch.arange(24 * 6, dtype=torch.float32).reshape(24, 1, 6)
    out = ext({"perception": torch.from_numpy(gold["x"]), "navigation": nav})
    assert out.shape == (24, 18)
    assert np.abs(out[:, :12].numpy() - gold["shallow_features"]).max() <= 1e-6
    assert torch.equal(out[:, 12:], nav[:, 0, :])
    twin = copy.deepcopy(ext)                     # policies are copied and pickled
    assert torch.equal(twin({"perception": torch.from_numpy(gold["x"]), "navigation": nav}), out)
    with pytest.raises(FileNotFoundError):
        compat.PerceptionNavigationExtractor(space)   # the reference's relative path, absent here


def test_radarcnn_import_line_works_after_install():
    compat.install_reference_names()
    from radarCNN import PerceptionNavigationExtractor, LidarCNN_shallow_pretrained   # noqa: F401  (radarCNN.py)
    assert PerceptionNavigationExtractor is compat.PerceptionNavigationExtractor
