"""The LiDAR-CNN consumers of the scan, on the GPU, reading the simulation's
perception buffer in place.

Mirrors the reference's model classes by name and head:

  kind        full network ("risk" head)                 RL feature extractor ("features" head)
  "shallow"   LidarCNN_shallow   models/lidar_cnn_shallow.py:8   LidarCNN_shallow_pretrained      radarCNN.py:148 (12)
  "deep2"     LidarCNN_2_deep    models/lidar_cnn_deep.py:98     LidarCNN_not_so_deep_pretrained  radarCNN.py:96  (12)
  "deep"      LidarCNN_deep      models/lidar_cnn_deep.py:10     LidarCNN_deep_pretrained         radarCNN.py:10  (8)

Weights are a state dict with the reference's own keys (what
`torch.load(".../model_shallow_pretrained.json")` returns, trainer.py:180-182).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

KINDS = {"shallow": _lib.CNN_SHALLOW, "deep2": _lib.CNN_2DEEP, "deep": _lib.CNN_DEEP}
HEADS = {"features": _lib.CNN_FEATURES, "risk": _lib.CNN_RISK}

# state-dict keys in state-dict order
STATE_KEYS = {
    "shallow": ("feature_extractor.0.weight", "feature_extractor.0.bias", "linear.0.weight", "linear.0.bias"),
    "deep2": ("feature_extractor.0.weight", "feature_extractor.0.bias", "feature_extractor.2.weight",
              "feature_extractor.2.bias", "feature_extractor.4.weight", "feature_extractor.4.bias",
              "linear_1.0.weight", "linear_1.0.bias"),
    "deep": ("feature_extractor.0.weight", "feature_extractor.0.bias", "feature_extractor.3.weight",
             "feature_extractor.3.bias", "feature_extractor.6.weight", "feature_extractor.6.bias",
             "feature_extractor.9.weight", "feature_extractor.9.bias", "linear_1.0.weight", "linear_1.0.bias",
             "linear_1.2.weight", "linear_1.2.bias", "linear_2.1.weight", "linear_2.1.bias"),
}


def pack_weights(kind: str, state_dict) -> np.ndarray:
    """Flatten a state dict (torch tensors or arrays) into the library's order."""
    parts = []
    for k in STATE_KEYS[kind]:
        v = state_dict[k]
        v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        parts.append(v.astype(np.float32).ravel())
    return np.ascontiguousarray(np.concatenate(parts))


def load_state_dict(path: str):
    """Read one of the reference's weight files (torch zip archives despite
    their .json suffix) without executing anything from it."""
    import torch

    return torch.load(path, weights_only=True, map_location="cpu")


class LidarCNN:
    def __init__(self, kind: str, state_dict, device: int = 0):
        self._lib = _lib.load()
        self.kind = kind
        self.device = int(device)
        w = pack_weights(kind, state_dict)
        self._h = C.c_void_p()
        rc = self._lib.auv_cnn_create(KINDS[kind], w.ctypes.data_as(C.c_void_p), w.size, self.device,
                                      C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise _lib.AuvError(rc, self._lib.auv_cnn_last_error(None).decode())

    def out_dim(self, head: str) -> int:
        return self._lib.auv_cnn_out_dim(KINDS[self.kind], HEADS[head])

    def forward(self, scans, head: str = "risk", out=None):
        """scans: float32 CUDA tensor (B,1,180), e.g. obs["perception"].
        Returns a (B, out_dim) float32 CUDA tensor; runs on the current stream."""
        import torch

        if not (scans.is_cuda and scans.dtype == torch.float32 and scans.is_contiguous()):
            raise ValueError("scans must be a contiguous float32 CUDA tensor")
        if scans.dim() != 3 or scans.shape[1] != 1 or scans.shape[2] != 180:
            raise ValueError(f"scans must be (B,1,180), got {tuple(scans.shape)}")
        if scans.device.index != self.device:
            raise ValueError(f"scans on cuda:{scans.device.index}, network on cuda:{self.device}")
        B = scans.shape[0]
        if out is None:
            out = torch.empty((B, self.out_dim(head)), dtype=torch.float32, device=scans.device)
        elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.device == scans.device
                  and tuple(out.shape) == (B, self.out_dim(head))):
            raise ValueError(f"out must be a contiguous float32 ({B}, {self.out_dim(head)}) tensor on {scans.device}")
        if B == 0:
            return out
        st = torch.cuda.current_stream(scans.device).cuda_stream
        rc = self._lib.auv_cnn_forward(self._h, C.c_void_p(scans.data_ptr()), B, HEADS[head],
                                       C.c_void_p(out.data_ptr()), C.c_void_p(st))
        if rc != 0:
            raise _lib.AuvError(rc, self._lib.auv_cnn_last_error(self._h).decode())
        return out

    __call__ = forward

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.auv_cnn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PerceptionNavigationExtractor:
    """What radarCNN.py:195-280 computes per policy step: the frozen CNN's
    features of obs["perception"] concatenated with obs["navigation"][:,0,:]."""

    def __init__(self, kind: str, state_dict, device: int = 0):
        self.cnn = LidarCNN(kind, state_dict, device)
        self.features_dim = self.cnn.out_dim("features") + 6

    def __call__(self, obs):
        import torch

        per = self.cnn.forward(obs["perception"], "features")
        nav = obs["navigation"][:, 0, :]
        return torch.cat([per, nav], dim=1)
