"""`VecAuvEnv`: the gym-style vectorised env over `AuvSim`.

Keeps the contract the reference's RL-side consumer reads
(`PerceptionNavigationExtractor`, radarCNN.py:195-280): `reset()` returns and
`step()` yields a dict observation with keys "perception" (B,1,180) float32
closeness and "navigation" (B,1,6) float32; `step()` is the old-gym 4-tuple
(obs, reward, done, info) that `import gym` + stable-baselines3
(radarCNN.py:1,6-7) expect.  The observation tensors alias device memory
(DLPack, zero-copy): they are overwritten by the next step.
"""
from __future__ import annotations

import numpy as np

from .params import NAV_FEATURES
from .sim import AuvSim


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


class DictSpace:
    """Minimal stand-in for gym.spaces.Dict: `.spaces` is what
    radarCNN.py:215 iterates."""

    def __init__(self, spaces):
        self.spaces = dict(spaces)


class VecAuvEnv:
    """auto_reset:
      True / "host"  a done env is reset right after the step that ended it, through
                     a host-side mask (one device -> host copy of the flags per step);
                     the step returns the fresh observation, `info["terminal_mask"]`
                     marks the envs that ended;
      "device"       no host round trip: a done env restarts INSIDE the next step
                     (its action is ignored, reward 0); the step that ends an episode
                     returns the terminal observation with done = True;
      False          done envs keep stepping until `reset()`.
    scene None: scenes, paths and initial states come from the device-side
    generator (`seed`, `gen_params`); with auto_reset="device" every restarting
    env then gets a new scene for its next episode, still without the host.
    """

    def __init__(self, cfg: dict, scene: dict | None, device: int = 0, auto_reset=True, seed: int = 0,
                 gen_params: dict | None = None, cnn=None, cnn_head: str = "features"):
        import torch

        from . import _lib

        self._torch = torch
        self.cfg = cfg
        self.scene = scene
        self.num_envs = int(cfg["n_envs"])
        self.device = torch.device("cuda", device)
        self.sim = AuvSim(cfg, device=device)
        if scene is None:
            self.sim.generate_scene(seed, gen_params)
        else:
            self.sim.set_scene(scene)
        self.auto_reset = "host" if auto_reset is True else auto_reset
        if self.auto_reset not in ("host", "device", False):
            raise ValueError("auto_reset must be True, 'host', 'device' or False")
        if self.auto_reset == "device":
            self.sim.set_option(_lib.OPT_AUTO_RESET, 1)
            if scene is None:
                self.sim.set_option(_lib.OPT_REGENERATE, 1)
        elif scene is None and self.auto_reset == "host":
            raise ValueError("generated scenes restart on the device: use auto_reset='device' or False")
        R = int(cfg["n_rays"])
        self.observation_space = DictSpace({
            "perception": Box(0.0, 1.0, (1, R)),
            "navigation": Box(-np.inf, np.inf, (1, NAV_FEATURES)),
        })
        self.action_space = Box(np.array([0.0, -1.0]), np.array([1.0, 1.0]), (2,))
        self._obs = {"perception": self.sim.tensor("perception"),
                     "navigation": self.sim.tensor("navigation")}
        # a `cnn.LidarCNN` given here is evaluated inside every sensing pass; its
        # output rides along as obs["cnn"] (B, out_dim): with the features head,
        # what PerceptionNavigationExtractor computes from obs["perception"]
        # (radarCNN.py:270-280) is then torch.cat([obs["cnn"], obs["navigation"][:, 0]], 1)
        if cnn is not None:
            self.sim.attach_cnn(cnn, cnn_head)
            self._obs["cnn"] = self.sim.tensor("cnn_out")
        self._reward = self.sim.tensor("reward")
        self._done = self.sim.tensor("done")
        self._action = self.sim.tensor("action")   # f64 [2,V] staging
        self._init_state = None if scene is None else np.ascontiguousarray(scene["init_state"], dtype=np.float64)
        self.episodes_done = 0

    def _stream(self):
        return self._torch.cuda.current_stream(self.device)

    def reset(self):
        """Reset every env; returns the observation dict."""
        self.sim.reset(self._init_state, stream=self._stream())
        return self._obs

    def step(self, actions):
        """actions: (B,2) torch tensor on the env's GPU or numpy array,
        columns (surge in [0,1], steer in [-1,1])."""
        torch = self._torch
        st = self._stream()
        if isinstance(actions, np.ndarray):
            a = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).T)
            self.sim.step_host(a, stream=st)
        else:
            if actions.shape != (self.num_envs, 2):
                raise ValueError(f"actions must be ({self.num_envs}, 2)")
            # transpose + widen into the library's [2,V] f64 staging buffer
            self._action.copy_(actions.to(self.device).t())
            self.sim.step_device(None, stream=st)
        info = {}
        if self.auto_reset == "host":
            done_host = self._done.cpu().numpy().astype(bool)
            if done_host.any():
                info["terminal_mask"] = done_host
                self.episodes_done += int(done_host.sum())
                reward = self._reward.clone()
                done = torch.from_numpy(done_host).to(self.device)
                self.sim.reset(self._init_state, mask=done_host, stream=st)
                return self._obs, reward, done, info
        return self._obs, self._reward, self._done.bool(), info

    def close(self):
        self.sim.close()
