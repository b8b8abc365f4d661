"""examples/pipeline.py end to end at a small size: generated scenes, rows logged on the device (and as CSV),
a network trained by the fused kernels, the trained network attached as the env's feature extractor."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,dim", [("deep2", 12), ("deep", 8)])
def test_pipeline_example(tmp_path, model, dim):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "pipeline.py")
    spec = importlib.util.spec_from_file_location("pipeline_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.run(envs=512, steps=150, model=model, epochs=3, csv_dir=str(tmp_path), verbose=False)
    assert out["rows"] > 2000 and out["rows_written"] == out["rows"]
    assert out["policy_input_shape"] == (512, dim + 6) and out["finite"]
    assert np.isfinite(out["test_mse"]) and out["train_loss"][-1] <= out["train_loss"][0]
    from gym_auv_amd import dataset as D

    data = D.load_lidar_dataset(os.path.join(str(tmp_path), "LiDAR_MovingObstaclesNoRules.csv"),
                                os.path.join(str(tmp_path), "risk_MovingObstaclesNoRules.csv"), mode="max")
    assert sum(len(data[k][0]) for k in ("train", "val", "test")) == out["rows"]
