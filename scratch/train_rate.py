"""Time per batch of the fused training kernels against eager torch on the same device
(random scans, batch 16, AdamW): python scratch/train_rate.py"""
import time

import numpy as np
import torch

from gym_auv_amd import trainer as T
from gym_auv_amd.models import MODELS

n = 4937 // 16 * 16
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (n, 1, 180)).astype(np.float32)
y = rng.uniform(0, 1, (n, 1)).astype(np.float32)
for kind in ("shallow", "deep2", "deep"):
    torch.manual_seed(1)
    dev = T.DeviceTrainer(MODELS[kind](), 1, 1e-3, X, y)
    dev.train_epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dev.train_epoch()
    torch.cuda.synchronize()
    fused = (time.perf_counter() - t0) / 3 / (n // 16)
    ref = T.Trainer(MODELS[kind](), 1, 1e-3, device="cuda")
    Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
    def eager(k):
        for i in range(k):
            ref.train_batch(Xd[i * 16:(i + 1) * 16], yd[i * 16:(i + 1) * 16])
    eager(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eager(100)
    torch.cuda.synchronize()
    eag = (time.perf_counter() - t0) / 100
    print(f"{kind}: fused {fused * 1e6:.1f} us/batch ({n // 16} batches per launch), eager torch {eag * 1e6:.0f} us/batch, x{eag / fused:.0f}", flush=True)
