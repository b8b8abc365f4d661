"""Regression trainer of the LiDAR-CNN risk models (mirror of the reference's
`Trainer`, trainer.py:18-90: MSE loss; AdamW, or SGD with momentum 0.9;
per-batch step, per-epoch validation)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


class Trainer:
    def __init__(self, model: nn.Module, epochs: int, learning_rate: float, optimizer: str = "adamw",
                 device: str = "cpu"):
        self.model = model.to(device)
        self.device = device
        self.epochs = int(epochs)
        self.loss = nn.MSELoss()
        if optimizer == "sgd":
            self.optimizer = torch.optim.SGD(self.model.parameters(), lr=learning_rate, momentum=0.9)
        else:
            self.optimizer = torch.optim.AdamW(self.model.parameters(), lr=learning_rate)
        self.training_loss, self.validation_loss = [], []

    def train_batch(self, X, y):
        self.model.train()
        loss = self.loss(self.model(X.to(self.device)), y.to(self.device))
        loss.backward()
        self.optimizer.step()
        self.optimizer.zero_grad()
        return float(loss.item())

    @torch.no_grad()
    def validation_step(self, batches):
        self.model.eval()
        losses = [float(self.loss(self.model(X.to(self.device)), y.to(self.device)).item()) for X, y in batches]
        return float(np.mean(losses)) if losses else float("nan")

    def train(self, train_batches, val_batches=()):
        """`train_batches` / `val_batches`: callables returning an iterable of
        (X, y) per epoch, or plain lists."""
        for _ in range(self.epochs):
            tb = train_batches() if callable(train_batches) else train_batches
            losses = [self.train_batch(X, y) for X, y in tb]
            self.training_loss.append(float(np.mean(losses)))
            vb = val_batches() if callable(val_batches) else val_batches
            self.validation_loss.append(self.validation_step(vb))
        return self.training_loss, self.validation_loss


def batches(X: np.ndarray, y: np.ndarray, batch_size: int = 16, shuffle: bool = True, drop_last: bool = True,
            seed: int = 2):
    """Mini-batches of a split from `dataset.load_lidar_dataset` (batch 16,
    drop_last, shuffled: trainer.py:120-125, utils/dataloader.py:76-80)."""
    rng = np.random.default_rng(seed)

    def one_epoch():
        idx = rng.permutation(len(X)) if shuffle else np.arange(len(X))
        end = len(X) - (len(X) % batch_size if drop_last else 0)
        for i in range(0, end, batch_size):
            j = idx[i:i + batch_size]
            yield torch.from_numpy(X[j]), torch.from_numpy(y[j])
    return one_epoch
