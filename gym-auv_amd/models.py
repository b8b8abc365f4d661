"""PyTorch modules of the three LiDAR-CNN risk models, for training.

Same layers, hyper-parameters and state-dict keys as the reference
(`models/lidar_cnn_shallow.py:23-40`, `models/lidar_cnn_deep.py:19-82` and
`:110-146`), so its weight files load with `strict=True` and a state dict
trained here feeds `gym_auv_amd.cnn.LidarCNN` (the HIP inference path)
unchanged.  Inference in the simulation loop does not use these modules.
"""
from __future__ import annotations

import torch.nn as nn


def _conv(cin, cout, k, stride, pad):
    return nn.Conv1d(cin, cout, kernel_size=k, stride=stride, padding=pad, padding_mode="circular")


class LidarCNNShallow(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature_extractor = nn.Sequential(_conv(1, 1, 45, 15, 15), nn.ReLU(), nn.Flatten())
        self.linear = nn.Sequential(nn.Linear(12, 1), nn.ReLU())

    def forward(self, x):
        return self.linear(self.feature_extractor(x))


class LidarCNN2Deep(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature_extractor = nn.Sequential(
            _conv(1, 3, 45, 15, 15), nn.ReLU(), _conv(3, 2, 3, 1, 1), nn.ReLU(), _conv(2, 1, 3, 1, 1), nn.Flatten())
        self.linear_1 = nn.Sequential(nn.Linear(12, 1), nn.ReLU())

    def forward(self, x):
        return self.linear_1(self.feature_extractor(x))


class LidarCNNDeep(nn.Module):
    """channels [2,4,4,6], kernel 9: the configuration of the shipped weights (trainer.py:144)."""

    def __init__(self, channels=(2, 4, 4, 6), kernel_size=9):
        super().__init__()
        layers, cin, pad = [], 1, (kernel_size - 1) // 2
        for c in channels:
            layers += [_conv(cin, c, kernel_size, 1, pad), nn.ReLU(), nn.MaxPool1d(2, 2, ceil_mode=True)]
            cin = c
        self.feature_extractor = nn.Sequential(*layers, nn.Flatten())
        self.linear_1 = nn.Sequential(nn.Linear(12 * channels[-1], 40), nn.ReLU(), nn.Linear(40, 8))
        self.linear_2 = nn.Sequential(nn.ReLU(), nn.Linear(8, 1), nn.ReLU())

    def forward(self, x):
        return self.linear_2(self.linear_1(self.feature_extractor(x)))


MODELS = {"shallow": LidarCNNShallow, "deep2": LidarCNN2Deep, "deep": LidarCNNDeep}
