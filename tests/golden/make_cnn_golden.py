"""Generate tests/golden/cnn_golden.npz from the REFERENCE's own model classes
and shipped weights.  Runs only where /root/reference is mounted; the .npz it
writes (weights as plain arrays, seeded inputs, expected outputs) is what
travels.  Weights are loaded with torch.load(weights_only=True).

Reference pieces exercised:
  models/lidar_cnn_shallow.py:8-52   LidarCNN_shallow      + logs/trained_models/model_shallow_pretrained.json
  models/lidar_cnn_deep.py:98-157    LidarCNN_2_deep       + model_2_deep_pretrained.json
  models/lidar_cnn_deep.py:10-96     LidarCNN_deep         + model_deep_pretrained.json
The RL-side extractors of radarCNN.py (:10-182) cannot be imported here (gym /
stable_baselines3 are absent); they are prefixes of the same networks with the
same state-dict keys, so their expected features are taken from the imported
models' sub-modules: shallow: feature_extractor[0] (conv, no ReLU) -> 12;
2_deep: feature_extractor -> 12; deep: feature_extractor + linear_1 -> 8.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from models.lidar_cnn_deep import LidarCNN_2_deep, LidarCNN_deep  # noqa: E402
from models.lidar_cnn_shallow import LidarCNN_shallow  # noqa: E402

W = os.path.join(REF, "logs", "trained_models")


def load(model, name):
    sd = torch.load(os.path.join(W, name), weights_only=True, map_location="cpu")
    model.load_state_dict(sd, strict=True)
    return model.eval()


def main():
    torch.manual_seed(0)
    shallow = load(LidarCNN_shallow(), "model_shallow_pretrained.json")
    deep2 = load(LidarCNN_2_deep(), "model_2_deep_pretrained.json")
    deep = load(LidarCNN_deep(n_sensors=180, output_channels=[2, 4, 4, 6], kernel_size=9), "model_deep_pretrained.json")
    rng = np.random.default_rng(0)
    x = rng.uniform(0.0, 1.0, (24, 1, 180)).astype(np.float32)
    x[0] = 0.0                                   # empty scan (closeness 0 everywhere)
    x[1] = 1.0                                   # touching everywhere
    x[2, 0, :] = 0.0
    x[2, 0, 170:] = 0.9                          # obstacle across the wrap-around
    x[2, 0, :10] = 0.9
    x[3, 0, :] = np.linspace(0, 1, 180, dtype=np.float32)
    # seeded check vector: torch.manual_seed(0); torch.rand(4,1,180)
    torch.manual_seed(0)
    x[4:8] = torch.rand(4, 1, 180).numpy()
    xt = torch.from_numpy(x)
    out = {"x": x}
    with torch.no_grad():
        out["shallow_risk"] = shallow(xt).numpy()
        out["shallow_features"] = shallow.feature_extractor[0](xt).flatten(1).numpy()
        out["deep2_risk"] = deep2(xt).numpy()
        f2 = xt
        for layer in deep2.feature_extractor:
            f2 = layer(f2)
        out["deep2_features"] = f2.numpy()
        out["deep_risk"] = deep(xt).numpy()
        fd = xt
        for layer in deep.feature_extractor:
            fd = layer(fd)
        for layer in deep.linear_1:
            fd = layer(fd)
        out["deep_features"] = fd.numpy()
    for tag, m in (("shallow", shallow), ("deep2", deep2), ("deep", deep)):
        for k, v in m.state_dict().items():
            out[f"w_{tag}/{k}"] = v.numpy()
    here = os.path.dirname(os.path.abspath(__file__))
    np.savez_compressed(os.path.join(here, "cnn_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})
    print("seeded check:", out["shallow_risk"][4:8, 0], out["deep2_risk"][4:8, 0], out["deep_risk"][4:8, 0])


if __name__ == "__main__":
    main()
