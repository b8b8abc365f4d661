"""Regression trainer of the LiDAR-CNN risk models (mirror of the reference's
`Trainer`, trainer.py:18-90: MSE loss; AdamW, or SGD with momentum 0.9;
per-batch step, per-epoch validation)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


class Trainer:
    def __init__(self, model: nn.Module, epochs: int, learning_rate: float, optimizer: str = "sgd",
                 device: str = "cpu"):
        """optimizer: 'adam' (or 'adamw') is AdamW, anything else SGD with momentum 0.9, and SGD is
        the default: the choice of the reference's constructor (trainer.py:25,35-38)."""
        self.model = model.to(device)
        self.device = device
        self.epochs = int(epochs)
        self.loss = nn.MSELoss()
        if optimizer in ("adam", "adamw"):
            self.optimizer = torch.optim.AdamW(self.model.parameters(), lr=learning_rate)
        else:
            self.optimizer = torch.optim.SGD(self.model.parameters(), lr=learning_rate, momentum=0.9)
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


class DeviceTrainer:
    """`Trainer` with the data resident on the GPU: every epoch's batches
    (trainer.py:80-83) run in one launch of the library's fused training
    kernel (`auv_train_cnn`: forward, MSE, backward and the optimizer step of
    trainer.py:44-53 per batch, parameters and optimizer state on chip),
    instead of 10 to 150 small launches per batch.

    X_* float32 [n,1,180] or [n,180] closeness scans and y_* float32 [n] or
    [n,1] labels, numpy or torch (copied to the device once).  `model` is one
    of `models.LidarCNNShallow`, `LidarCNN2Deep` or `LidarCNNDeep` (with the
    default channels and kernel, the configuration of trainer.py:144); its
    parameters are read at construction and written back after every epoch,
    so `model.state_dict()` feeds `cnn.LidarCNN` (the HIP inference path) as
    before."""

    def __init__(self, model, epochs: int, learning_rate: float, X_train, y_train, X_val=None, y_val=None,
                 optimizer: str = "sgd", batch_size: int = 16, seed: int = 2, device: int = 0):
        import ctypes as C

        from . import _lib
        from .cnn import KINDS, STATE_KEYS, pack_weights
        from .models import MODELS

        self._lib, self._C, self._api = _lib.load(), C, _lib
        if not 1 <= batch_size <= _lib.TRAIN_BATCH_MAX:
            raise ValueError(f"batch_size must be in 1..{_lib.TRAIN_BATCH_MAX}")
        kinds = [k for k, cls in MODELS.items() if isinstance(model, cls)]   # subclasses: compat's reference-named ones
        if not kinds:
            raise TypeError(f"DeviceTrainer trains {', '.join(c.__name__ for c in MODELS.values())}; got {type(model).__name__}")
        self.kind = kinds[0]
        self._kind_id = KINDS[self.kind]
        self.model, self.epochs, self.device = model, int(epochs), int(device)
        self.batch_size = int(batch_size)
        dev = torch.device("cuda", self.device)
        as_dev = lambda t, shape: torch.as_tensor(np.asarray(t) if not torch.is_tensor(t) else t, dtype=torch.float32) \
            .reshape(shape).contiguous().to(dev)
        self.X, self.y = as_dev(X_train, (-1, 180)), as_dev(y_train, (-1,))
        self.Xv = None if X_val is None else as_dev(X_val, (-1, 180))
        self.yv = None if y_val is None else as_dev(y_val, (-1,))
        if len(self.X) != len(self.y):
            raise ValueError("X_train and y_train differ in length")
        self._keys = STATE_KEYS[self.kind]
        packed = pack_weights(self.kind, model.state_dict())
        if packed.size != self._lib.auv_cnn_n_weights(self._kind_id):
            raise ValueError(f"{type(model).__name__} has {packed.size} parameters; the library's {self.kind} network has "
                             f"{self._lib.auv_cnn_n_weights(self._kind_id)} (default channels and kernel only)")
        self.w = torch.from_numpy(packed).to(dev)
        self.opt_state = torch.zeros(2 * self.w.numel(), dtype=torch.float32, device=dev)
        self.hp = _lib.AuvTrainParams(optimizer=_lib.OPT_ADAMW if optimizer in ("adam", "adamw") else _lib.OPT_SGD_MOMENTUM,
                                      reserved=0, lr=float(learning_rate), beta1=0.9, beta2=0.999, eps=1e-8,
                                      weight_decay=0.01, momentum=0.9)
        self.steps_done = 0
        self._gen = torch.Generator().manual_seed(int(seed))
        self.training_loss, self.validation_loss = [], []

    def train_epoch(self, perm=None):
        """One pass over the training rows in batches (drop_last); `perm`: the row
        order (default: a fresh shuffle).  Returns the per-batch losses (device tensor)."""
        C = self._C
        n = len(self.X)
        if perm is None:
            perm = torch.randperm(n, generator=self._gen)
        n_batches = n // self.batch_size
        perm = torch.as_tensor(perm)
        if perm.numel() < n_batches * self.batch_size:
            raise ValueError(f"perm holds {perm.numel()} rows; {n_batches} batches of {self.batch_size} need {n_batches * self.batch_size}")
        perm = perm[:n_batches * self.batch_size]
        if n_batches and (int(perm.min()) < 0 or int(perm.max()) >= n):     # the kernel indexes X and y with it unchecked
            raise ValueError(f"perm must hold row numbers in 0..{n - 1}")
        perm_d = perm.to(torch.int32).contiguous().to(self.X.device)
        losses = torch.empty(n_batches, dtype=torch.float32, device=self.X.device)
        st = torch.cuda.current_stream(self.X.device).cuda_stream
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = self._lib.auv_train_cnn(self._kind_id, self.device, p(self.X), p(self.y), n, p(perm_d), n_batches,
                                     self.batch_size, p(self.w), p(self.opt_state), C.byref(self.hp), self.steps_done,
                                     p(losses), C.c_void_p(st))
        if rc != 0:
            raise self._api.AuvError(rc, self._lib.auv_cnn_last_error(None).decode())
        self.steps_done += n_batches
        return losses

    def sync_model(self):
        """Write the trained parameters back into `model`."""
        flat = self.w.detach().cpu()
        sd, o = {}, 0
        ref = self.model.state_dict()
        for k in self._keys:
            n = ref[k].numel()
            sd[k] = flat[o:o + n].reshape(ref[k].shape).clone()
            o += n
        self.model.load_state_dict(sd)
        return self.model

    @torch.no_grad()
    def validation_step(self, perm=None):
        """`perm`: the order in which a shuffling validation loader hands out the
        rows (which rows its dropped partial batch leaves out depends on it);
        default: file order."""
        if self.Xv is None or len(self.Xv) == 0:
            return float("nan")
        from .cnn import LidarCNN

        net = LidarCNN(self.kind, self.sync_model().state_dict(), device=self.device)
        pred = net.forward(self.Xv.reshape(-1, 1, 180), "risk")[:, 0]
        net.close()
        # the reference averages the losses of the validation batches (trainer.py:55-69), and its validation
        # loader drops the last partial batch (utils/dataloader.py:86-90): full batches only
        b = self.batch_size
        per = (pred - self.yv) ** 2
        if perm is not None:
            per = per[torch.as_tensor(perm).to(per.device)]
        n_full = len(per) // b
        if n_full == 0:
            return float("nan")
        return float(per[:n_full * b].reshape(n_full, b).mean(dim=1).mean().item())

    def train(self):
        for _ in range(self.epochs):
            self.training_loss.append(float(self.train_epoch().mean().item()))
            self.validation_loss.append(self.validation_step())
        self.sync_model()
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
