"""The reference's whole workflow on one MI355X, nothing but text files ever touching the host:

  1. a vectorised env over generated MovingObstaclesNoRules-like scenes (episodes restart on the device),
  2. the LiDAR-risk dataset from the running env: rows listed by the reference's rule, kept on the device as
     training samples and (optionally) written as the three CSV files the reference's loader reads,
  3. one of the reference's risk networks trained on those rows by the fused training kernels
     (trainer.py:18-90: MSE, AdamW, batch 16, sequential split of utils/dataloader.py:63-67),
  4. the trained network attached to the env as the frozen feature extractor of the RL side
     (radarCNN.py:195-280): obs["cnn"] joined with obs["navigation"].

    PYTHONPATH=. python examples/pipeline.py [--envs 4096] [--steps 400] [--model deep] [--csv DIR]
"""
from __future__ import annotations

import argparse
import time

import numpy as np
import torch

from gym_auv_amd import cnn, dataset, params, scenarios
from gym_auv_amd.env import VecAuvEnv
from gym_auv_amd.models import MODELS
from gym_auv_amd.trainer import DeviceTrainer


def helmsman(n, rng, device):
    """Commands held for a while: mostly ahead at 60-100 % thrust, small rudder."""
    a = np.stack([rng.uniform(0.6, 1.0, n), rng.uniform(-0.3, 0.3, n)], axis=1)
    return torch.from_numpy(a).to(device)


def run(envs=4096, steps=400, model="shallow", epochs=8, lr=2e-3, csv_dir=None, seed=0, verbose=True):
    say = print if verbose else (lambda *a, **k: None)
    rng = np.random.default_rng(seed)
    cfg = params.make_config(envs, 8, 8, 5, max_steps=400)
    env = VecAuvEnv(cfg, None, auto_reset="device", seed=seed, gen_params=scenarios.REFERENCE_LIKE)
    obs = env.reset()
    writer = dataset.DatasetWriter(csv_dir) if csv_dir else None
    logger = dataset.DeviceDatasetLogger(env.sim, writer, collect="max")

    # 1 + 2: simulate and log
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    actions = helmsman(envs, rng, env.device)
    for k in range(steps):
        if k % 10 == 0:
            actions = helmsman(envs, rng, env.device)
        obs, reward, done, info = env.step(actions)
        logger.after_step()
    torch.cuda.synchronize()
    t_sim = time.perf_counter() - t0
    if writer:
        writer.close()
    X, y = logger.samples()
    say(f"simulated {envs} vessels x {steps} steps and logged {len(X)} rows in {t_sim:.2f} s "
        f"({envs * steps / t_sim:.3g} env-steps/s with the logging, {len(X) / t_sim:.3g} rows/s"
        f"{', CSV files in ' + csv_dir if csv_dir else ''})")

    # 3: train on the rows, the reference's sequential split
    n_train, n_val, n_test = dataset.split_sizes(len(X))
    tr, va, te = slice(0, n_train), slice(n_train, n_train + n_val), slice(len(X) - n_test, len(X))
    torch.manual_seed(2)
    net = MODELS[model]()
    trainer = DeviceTrainer(net, epochs=epochs, learning_rate=lr, X_train=X[tr], y_train=y[tr], X_val=X[va], y_val=y[va],
                            optimizer="adam")   # the reference's own run (trainer.py:144-149)
    t0 = time.perf_counter()
    train_loss, val_loss = trainer.train()
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    frozen = cnn.LidarCNN(model, net.state_dict())
    pred = frozen.forward(X[te].contiguous(), "risk")
    test_mse = float(((pred - y[te]) ** 2).mean())
    label_var = float(y[te].var())
    say(f"trained {model} for {epochs} epochs ({trainer.steps_done} steps) in {t_train:.2f} s: "
        f"train loss {train_loss[0]:.4f} -> {train_loss[-1]:.4f}, validation {val_loss[-1]:.4f}, "
        f"test MSE {test_mse:.4f} (variance of the test labels {label_var:.4f})")

    # 4: the trained network as the env's frozen feature extractor
    env.sim.attach_cnn(frozen, "features")
    feats = env.sim.tensor("cnn_out")
    obs, reward, done, info = env.step(actions)
    policy_input = torch.cat([feats, obs["navigation"][:, 0, :]], dim=1)
    say(f"policy input per vessel: {tuple(policy_input.shape)} = {feats.shape[1]} LiDAR features + 6 navigation")
    out = {"rows": len(X), "test_mse": test_mse, "label_var": label_var, "train_loss": train_loss, "val_loss": val_loss,
           "policy_input_shape": tuple(policy_input.shape), "finite": bool(torch.isfinite(policy_input).all()),
           "rows_written": writer.rows if writer else 0}
    frozen.close()
    env.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--model", default="shallow", choices=sorted(MODELS))
    ap.add_argument("--epochs", type=int, default=8)
    ap.add_argument("--csv", default=None, help="also write LiDAR_/risk_/metadata_*.csv into this directory")
    a = ap.parse_args()
    run(a.envs, a.steps, a.model, a.epochs, csv_dir=a.csv)
