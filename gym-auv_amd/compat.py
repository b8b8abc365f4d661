"""The reference's own names and call signatures for the consumer side of the
hot path, so that its training script runs against this package with only its
import lines changed:

    from gym_auv_amd.compat import LidarCNN_deep, LidarCNN_2_deep      # models/lidar_cnn_deep.py
    from gym_auv_amd.compat import LidarCNN_shallow                    # models/lidar_cnn_shallow.py
    from gym_auv_amd.compat import load_LiDARDataset                   # utils/dataloader.py
    from gym_auv_amd.compat import Trainer                             # trainer.py:18

or, with the import lines left as they are, after `compat.install_reference_names()`
(which registers `models.lidar_cnn_deep`, `models.lidar_cnn_shallow` and
`utils.dataloader` as importable modules holding these objects).

The models are torch modules with the reference's layers and state-dict keys
(`gym_auv_amd.models`), the loader returns the reference's 6-tuple, and
`Trainer` has the reference's constructor; on a machine with a GPU it runs the
epochs through the library's fused training kernel (`trainer.DeviceTrainer`)
when the model is one of the three shipped configurations, with the very
batches the data loaders would have produced.

The RL side (radarCNN.py) is here under its own names too:
`LidarCNN_shallow_pretrained`, `LidarCNN_not_so_deep_pretrained`,
`LidarCNN_deep_pretrained`, `NavigatioNN` and `PerceptionNavigationExtractor`,
feature extractors in the sense of stable-baselines3 (its `BaseFeaturesExtractor`
is their base class when that package is importable) with the reference's
constructors and state-dict keys.  Frozen, on a GPU and in their shipped
configuration they evaluate through the library's CNN kernel, reading the
simulation's perception buffer in place.
"""
from __future__ import annotations

import sys
import types

import numpy as np
import torch

from . import dataset as _dataset
from . import models as _models
from . import trainer as _trainer

N_SENSORS = _dataset.N_SENSORS   # utils/dataloader.py:13


# ----------------------------------------------------------------------------
# models (models/lidar_cnn_shallow.py:8-52, models/lidar_cnn_deep.py:10-157)
# ----------------------------------------------------------------------------
class LidarCNN_shallow(_models.LidarCNNShallow):
    def __init__(self, n_sensors: int = 180, output_channels=(1,), kernel_size: int = 45, padding: int = 15,
                 stride: int = 15):
        super().__init__(tuple(output_channels), kernel_size, padding, stride)
        self.n_sensors, self.output_channels = n_sensors, list(output_channels)
        self.kernel_size, self.padding, self.stride = kernel_size, padding, stride


class LidarCNN_2_deep(_models.LidarCNN2Deep):
    def __init__(self, n_sensors: int = 180, output_channels=(3, 2, 1), kernel_size: int = 45):
        super().__init__(tuple(output_channels))   # kernel_size is stored and otherwise unused, as in the reference
        self.n_sensors, self.output_channels, self.kernel_size = n_sensors, list(output_channels), kernel_size


class LidarCNN_deep(_models.LidarCNNDeep):
    def __init__(self, n_sensors: int, output_channels, kernel_size: int = 5):
        super().__init__(tuple(output_channels), kernel_size, n_sensors)
        self.n_sensors, self.output_channels, self.kernel_size = n_sensors, list(output_channels), kernel_size
        self.padding = (kernel_size - 1) // 2


# ----------------------------------------------------------------------------
# data (utils/dataloader.py:15-101)
# ----------------------------------------------------------------------------
class LiDARDataset(torch.utils.data.Dataset):
    """Scans with a channel axis and labels as float32 tensors `.X`, `.y`
    (utils/dataloader.py:15-41); `prev_steps` stacks that many earlier scans
    behind each scan first."""

    def __init__(self, X, y, x_mean=0.0, x_std=1.0, prev_steps=None, standarize: bool = False):
        if not torch.is_tensor(X) and not torch.is_tensor(y) and standarize:
            X = (X - x_mean) / x_std
        if prev_steps:
            X = _dataset.prev_timesteps_feature_engineering(X, prev_steps)
        self.X = torch.as_tensor(np.asarray(X)[:, None], dtype=torch.float32)
        self.y = torch.as_tensor(np.asarray(y)[:, None], dtype=torch.float32)

    def __len__(self):
        return len(self.X)

    def __getitem__(self, i):
        return self.X[i], self.y[i]


def load_LiDARDataset(path_x: str, path_y: str, batch_size: int, mode: str = None, prev_steps=None,
                      train_test_split: float = 0.7, train_val_split: float = 0.2, shuffle: bool = True):
    """(data_train, data_val, data_test, dataloader_train, dataloader_val,
    dataloader_test) of the reference's loader (utils/dataloader.py:44-101):
    closeness 1 - X/150, labels reduced by `mode`, the sequential split, and
    torch data loaders of `batch_size` (1 for the test split) that drop the
    last partial batch.  The loaders read in this process (the reference asks
    for one worker process each; the batches are the same)."""
    X = 1.0 - np.loadtxt(path_x, ndmin=2) / _dataset.SENSOR_RANGE
    y = np.loadtxt(path_y) if mode is None else _dataset.calculate_total_risk(path_y, mode)
    n_train, n_val, n_test = _dataset.split_sizes(len(X), train_test_split, train_val_split)
    X_train = X[:n_train]
    x_mean, x_std = X_train.mean(), X_train.std()
    parts = [(X_train, y[:n_train], batch_size), (X[n_train:n_train + n_val], y[n_train:n_train + n_val], batch_size),
             (X[len(X) - n_test:], y[len(X) - n_test:], 1)]
    sets = [LiDARDataset(a, b, x_mean, x_std, prev_steps=prev_steps) for a, b, _ in parts]
    loaders = [torch.utils.data.DataLoader(d, batch_size=bs, shuffle=shuffle, num_workers=0, drop_last=True)
               for d, (_, _, bs) in zip(sets, parts)]
    return (*sets, *loaders)


calculate_total_risk = _dataset.calculate_total_risk                        # utils/dataloader.py:104
prev_timesteps_feature_enginering = _dataset.prev_timesteps_feature_engineering   # utils/dataloader.py:123 (its spelling)


def epoch_order(loader) -> torch.Tensor:
    """Row numbers of one pass over `loader`, batch after batch, drawn exactly
    as iterating the loader would draw them (same use of torch's global
    generator), without loading a row."""
    # a loader's iterator first takes a seed for its workers from the generator, used or not, then the sampler
    # draws the permutation
    torch.empty((), dtype=torch.int64).random_(generator=loader.generator)
    return torch.tensor([i for batch in loader.batch_sampler for i in batch], dtype=torch.int64)


# ----------------------------------------------------------------------------
# training (trainer.py:18-90)
# ----------------------------------------------------------------------------
class Trainer:
    """`Trainer(model, epochs, learning_rate, dataloader_train, dataloader_val,
    optimizer='sgd')` as in trainer.py:18-41: MSE loss; 'adam' is AdamW,
    anything else SGD with momentum 0.9.  `train()` fills `training_loss` and
    `validation_loss` per epoch.

    device: "auto" runs on the GPU through the fused training kernel when
    CUDA is present, the model is one of the three shipped configurations and
    the loaders are over `LiDARDataset`s of plain 180-ray scans with batches of
    at most 16; "cpu" is the reference's eager loop; "cuda" insists on the
    kernel.  Either way the batches are the ones the loaders yield."""

    def __init__(self, model, epochs: int, learning_rate: float, dataloader_train, dataloader_val,
                 optimizer: str = "sgd", device: str = "auto"):
        self.model, self.epochs, self.learning_rate = model, int(epochs), learning_rate
        self.dataloader_train, self.dataloader_val = dataloader_train, dataloader_val
        self.loss = torch.nn.MSELoss()
        self.training_loss, self.validation_loss = [], []
        self._dev = None
        why = self._device_obstacle() if device != "cpu" else "device='cpu'"
        if why is None:
            dt, dv = dataloader_train.dataset, dataloader_val.dataset
            self._dev = _trainer.DeviceTrainer(model, epochs, learning_rate, dt.X, dt.y, dv.X, dv.y, optimizer=optimizer,
                                               batch_size=dataloader_train.batch_size)
            self.optimizer = None
        else:
            if device == "cuda":
                raise ValueError(f"Trainer(device='cuda'): {why}")
            self._cpu = _trainer.Trainer(model, epochs, learning_rate, optimizer=optimizer, device="cpu")
            self.optimizer = self._cpu.optimizer
        self.on_device = self._dev is not None

    def _device_obstacle(self):
        """None when the fused kernel can run this training, else the reason it cannot."""
        from . import _lib

        if not torch.cuda.is_available():
            return "no GPU"
        if not any(isinstance(self.model, c) for c in _models.MODELS.values()):
            return f"{type(self.model).__name__} is not one of the three LiDAR networks"
        for ld in (self.dataloader_train, self.dataloader_val):
            ds = getattr(ld, "dataset", None)
            if not isinstance(ds, LiDARDataset) or ds.X.dim() != 3 or ds.X.shape[1:] != (1, N_SENSORS):
                return "the loaders are not over LiDARDatasets of [n,1,180] scans"
            if ld.batch_size is None or not ld.drop_last:
                return "the loaders do not drop the last partial batch"
        if not 1 <= self.dataloader_train.batch_size <= _lib.TRAIN_BATCH_MAX:
            return f"batch size above {_lib.TRAIN_BATCH_MAX}"
        if self.dataloader_val.batch_size != self.dataloader_train.batch_size:
            return "training and validation batch sizes differ"
        kind = [k for k, c in _models.MODELS.items() if isinstance(self.model, c)][0]
        n = sum(p.numel() for p in self.model.parameters())
        if n != _lib.load().auv_cnn_n_weights({"shallow": _lib.CNN_SHALLOW, "deep2": _lib.CNN_2DEEP, "deep": _lib.CNN_DEEP}[kind]):
            return "the model is not in its shipped configuration (channels / kernel)"
        return None

    def train_batch(self, X_batch, Y_batch) -> float:
        """One optimizer step on one batch, eagerly (trainer.py:44-53)."""
        if self._dev is not None:
            raise RuntimeError("this Trainer runs whole epochs on the GPU; build it with device='cpu' to step by batch")
        return self._cpu.train_batch(X_batch, Y_batch)

    def validation_step(self) -> float:
        """Mean of the validation batches' losses (trainer.py:55-69)."""
        if self._dev is not None:
            return self._dev.validation_step(epoch_order(self.dataloader_val))
        self.model.eval()
        v = self._cpu.validation_step(self.dataloader_val)
        self.model.train()
        return v

    def train(self):
        print("---------")   # the reference's banner (trainer.py:73-75)
        print("Training")
        print("---------")
        for e in range(self.epochs):
            if self._dev is not None:
                losses = self._dev.train_epoch(epoch_order(self.dataloader_train))
                train_loss = float(losses.double().sum().item()) / max(len(losses), 1)
            else:
                per = [self._cpu.train_batch(X, y) for X, y in self.dataloader_train]
                train_loss = sum(per) / len(per)
            val_loss = self.validation_step()
            self.training_loss.append(train_loss)
            self.validation_loss.append(val_loss)
            print("EPOCH", e + 1, ": \tTraining loss:", train_loss, "\tValidation loss:", val_loss)
        if self._dev is not None:
            self._dev.sync_model()


# ----------------------------------------------------------------------------
# RL-side feature extractors (radarCNN.py:10-280)
# ----------------------------------------------------------------------------
try:   # stable-baselines3 is what the reference derives them from (radarCNN.py:7)
    from stable_baselines3.common.torch_layers import BaseFeaturesExtractor as _ExtractorBase
except ImportError:
    class _ExtractorBase(torch.nn.Module):
        """What BaseFeaturesExtractor gives its subclasses: the observation
        space and the `features_dim` property."""

        def __init__(self, observation_space, features_dim: int = 0):
            super().__init__()
            assert features_dim > 0
            self._observation_space = observation_space
            self._features_dim = features_dim

        @property
        def features_dim(self) -> int:
            return self._features_dim


def _circ_conv(cin, cout, k, stride, pad):
    return torch.nn.Conv1d(cin, cout, k, stride=stride, padding=pad, padding_mode="circular")


class _ScanExtractor(_ExtractorBase):
    """Common part of the three: `forward` runs the layers of
    `feature_extractor` (and `linear_1`) in turn, as radarCNN.py:86-94,141-146,
    177-182 do; a frozen network in its shipped configuration, given a float32
    CUDA batch of [B,1,180] scans, is evaluated by the library's kernel instead
    (same values to rounding: tests/test_compat_gpu.py)."""

    _kind = None     # "shallow" | "deep2" | "deep": the library's network of which this is the feature part
    _shapes = ()     # parameter shapes of the shipped configuration, in state-dict order

    def _layers(self):
        yield from self.feature_extractor
        if hasattr(self, "linear_1"):
            yield from self.linear_1

    def _device_net(self, x):
        """The library's network for these weights on x's device, or None when
        the kernel does not apply."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and tuple(x.shape[1:]) == (1, N_SENSORS)):
            return None
        params = list(self.parameters())
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return None   # someone wants gradients: the torch layers
        if tuple(tuple(p.shape) for p in params) != self._shapes:
            return None
        from . import cnn as _cnn

        stamp = (x.device.index,) + tuple((p.data_ptr(), p._version) for p in params)
        if getattr(self, "_net_stamp", None) != stamp:
            old = getattr(self, "_net", None)
            if old is not None:
                old.close()
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            for k in _cnn.STATE_KEYS[self._kind]:   # the risk head's layers, which the features do not pass through
                if k not in sd:
                    n_in = 8 if self._kind == "deep" else 12
                    sd[k] = torch.zeros((1, n_in) if k.endswith("weight") else (1,))
            self.__dict__["_net"] = _cnn.LidarCNN(self._kind, sd, device=x.device.index)
            self.__dict__["_net_stamp"] = stamp
        return self.__dict__["_net"]

    def __getstate__(self):
        # copies and pickles of the module are made without the library's handle; the copy makes its own
        state = dict(self.__dict__)
        state.pop("_net", None)
        state.pop("_net_stamp", None)
        return state

    def forward(self, x):
        net = self._device_net(x)
        if net is not None:
            return net.forward(x.contiguous(), "features")
        for layer in self._layers():
            x = layer(x)
        return x


class LidarCNN_deep_pretrained(_ScanExtractor):
    """radarCNN.py:10-94: four circular convolutions with ReLU and ceil-mode
    max-pooling, then Linear(72, 40), ReLU, Linear(40, 8)."""
    _kind = "deep"
    _shapes = ((2, 1, 9), (2,), (4, 2, 9), (4,), (4, 4, 9), (4,), (6, 4, 9), (6,), (40, 72), (40,), (8, 40), (8,))

    def __init__(self, observation_space, output_channels=(2, 4, 4, 6), kernel_size: int = 9, n_sensors: int = 180,
                 features_dim: int = 8):
        super().__init__(observation_space, features_dim=features_dim)
        self.n_sensors, self.kernel_size = n_sensors, kernel_size
        self.padding = (kernel_size - 1) // 2
        self.output_channels = list(output_channels)
        layers, cin = [], 1
        for cout in self.output_channels[:4]:
            layers += [_circ_conv(cin, cout, kernel_size, 1, self.padding), torch.nn.ReLU(),
                       torch.nn.MaxPool1d(kernel_size=2, stride=2, ceil_mode=True)]
            cin = cout
        self.feature_extractor = torch.nn.Sequential(*layers, torch.nn.Flatten())
        self.linear_1 = torch.nn.Sequential(torch.nn.Linear(12 * self.output_channels[-1], 40), torch.nn.ReLU(),
                                            torch.nn.Linear(40, 8))


class LidarCNN_not_so_deep_pretrained(_ScanExtractor):
    """radarCNN.py:96-146: Conv1d(1 -> c0, 45, stride 15, circular padding 15),
    ReLU, two circular 3-tap convolutions (ReLU between), flattened."""
    _kind = "deep2"
    _shapes = ((3, 1, 45), (3,), (2, 3, 3), (2,), (1, 2, 3), (1,))

    def __init__(self, observation_space, n_sensors: int = 180, output_channels=(3, 2, 1), kernel_size: int = 45,
                 features_dim: int = 8):
        super().__init__(observation_space, features_dim=features_dim)
        self.n_sensors, self.kernel_size, self.output_channels = n_sensors, kernel_size, list(output_channels)
        c = self.output_channels
        self.feature_extractor = torch.nn.Sequential(
            _circ_conv(1, c[0], 45, 15, 15), torch.nn.ReLU(), _circ_conv(c[0], c[1], 3, 1, 1), torch.nn.ReLU(),
            _circ_conv(c[1], c[2], 3, 1, 1), torch.nn.Flatten())


class LidarCNN_shallow_pretrained(_ScanExtractor):
    """radarCNN.py:148-182: the one circular convolution, flattened (no ReLU,
    no linear layer): 12 features."""
    _kind = "shallow"
    _shapes = ((1, 1, 45), (1,))

    def __init__(self, observation_space, n_sensors: int = 180, output_channels=(1,), kernel_size: int = 45,
                 padding: int = 15, stride: int = 15, features_dim: int = 12):
        super().__init__(observation_space, features_dim=features_dim)
        self.n_sensors, self.kernel_size, self.padding, self.stride = n_sensors, kernel_size, padding, stride
        self.output_channels = list(output_channels)
        self.feature_extractor = torch.nn.Sequential(
            _circ_conv(1, self.output_channels[0], kernel_size, stride, padding), torch.nn.Flatten())


class NavigatioNN(_ExtractorBase):
    """radarCNN.py:184-193: the navigation features [B,1,n] passed through as [B,n]."""

    def __init__(self, observation_space, features_dim: int = 6):
        super().__init__(observation_space, features_dim=features_dim)
        self.passthrough = torch.nn.Identity()

    def forward(self, observations):
        shape = observations.shape
        return self.passthrough(observations[:, 0, :].reshape(shape[0], shape[-1]))


class PerceptionNavigationExtractor(_ExtractorBase):
    """radarCNN.py:195-280: per key of the Dict observation space an extractor
    -- "perception": the shallow network with the weights of `cnn_path` (the
    keys the file shares with it, radarCNN.py:243-250), frozen; "navigation":
    the pass-through -- and their outputs concatenated.  `cnn_path` (not a
    parameter of the reference, which has the path in its body) defaults to the
    reference's; the file is read without executing anything from it."""

    def __init__(self, observation_space, sensor_dim: int = 180, features_dim: int = 12, kernel_overlap: float = 0.05,
                 cnn_path: str = "gym_auv/utils/model_shallow_pretrained.json"):
        super().__init__(observation_space, features_dim=1)   # the real size is known after the loop
        extractors, total = {}, 0
        for key, subspace in observation_space.spaces.items():
            if key == "perception":
                net = LidarCNN_shallow_pretrained(observation_space=subspace, n_sensors=sensor_dim, output_channels=[1],
                                                  kernel_size=45, padding=15, stride=15, features_dim=features_dim)
                have = net.state_dict()
                pretrained = {k: v for k, v in torch.load(cnn_path, weights_only=True, map_location="cpu").items()
                              if k in have}
                print("\nLoading pretrained LidarCNN from", cnn_path)
                print("Pretraind layers:", [k for k in pretrained.keys()])
                net.load_state_dict(pretrained)
                for param in net.parameters():
                    param.requires_grad = False
                extractors[key] = net
                total += features_dim
            elif key == "navigation":
                extractors[key] = NavigatioNN(subspace, features_dim=subspace.shape[-1])
                total += subspace.shape[-1]
        self.extractors = torch.nn.ModuleDict(extractors)
        self._features_dim = total

    def forward(self, observations):
        return torch.cat([extractor(observations[key]) for key, extractor in self.extractors.items()], dim=1)


def install_reference_names():
    """Register `models.lidar_cnn_shallow`, `models.lidar_cnn_deep`,
    `utils.dataloader` and `radarCNN` in `sys.modules`, holding the objects
    above, so that a script written against the reference's layout imports them
    unchanged (trainer.py:6-7,11).  Refuses to shadow modules of those names that are
    already imported from somewhere else."""
    def module(name, **objs):
        have = sys.modules.get(name)
        if have is not None and not getattr(have, "__gym_auv_amd_compat__", False):
            raise RuntimeError(f"a module named {name!r} is already imported from {getattr(have, '__file__', '?')}")
        m = have or types.ModuleType(name)
        m.__gym_auv_amd_compat__ = True
        m.__dict__.update(objs)
        sys.modules[name] = m
        return m

    models = module("models")
    models.__path__ = []
    models.lidar_cnn_shallow = module("models.lidar_cnn_shallow", LidarCNN_shallow=LidarCNN_shallow)
    models.lidar_cnn_deep = module("models.lidar_cnn_deep", LidarCNN_deep=LidarCNN_deep, LidarCNN_2_deep=LidarCNN_2_deep)
    utils = module("utils")
    utils.__path__ = []
    module("radarCNN", LidarCNN_deep_pretrained=LidarCNN_deep_pretrained,
           LidarCNN_not_so_deep_pretrained=LidarCNN_not_so_deep_pretrained,
           LidarCNN_shallow_pretrained=LidarCNN_shallow_pretrained, NavigatioNN=NavigatioNN,
           PerceptionNavigationExtractor=PerceptionNavigationExtractor)
    utils.dataloader = module("utils.dataloader", load_LiDARDataset=load_LiDARDataset, LiDARDataset=LiDARDataset,
                              calculate_total_risk=calculate_total_risk, N_SENSORS=N_SENSORS,
                              prev_timesteps_feature_enginering=prev_timesteps_feature_enginering)
