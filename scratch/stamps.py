"""Where the waves of k_sense spend their time: mean shader-clock ticks between the phase stamps of a
diagnostic build (-DAUV_PHASE_STAMPS), per wave of the workgroup.  Usage: AUVSIM_LIB=<stamped lib> python scratch/stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_auv_amd import _lib, scenarios
from gym_auv_amd.sim import AuvSim

cfg, scene = scenarios.make_workload("cfg3_65536_moving")
cfg["risk_model"] = _lib.RISK_REFERENCE
sim = AuvSim(cfg, device=0)
sim.set_option(_lib.OPT_AUTO_RESET, 1)
sim.set_scene(scene)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev)
sim.reset(scene["init_state"], stream=stream)
V = cfg["n_envs"]
acts = torch.from_numpy(scenarios.random_actions(V, 8, seed=100)).to(dev)
gold = np.load(os.path.join(ROOT, "tests", "golden", "cnn_golden.npz"))
from gym_auv_amd import cnn
net = cnn.LidarCNN("shallow", {k[len("w_shallow/"):]: gold[k] for k in gold.files if k.startswith("w_shallow/")}, device=0)
sim.attach_cnn(net, "features")
blocks = V // 4
buf = torch.zeros((blocks, 4, 16), dtype=torch.int64, device=dev)
lib = _lib.load()
fn = lib.auv_debug_set_stamps
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
for k in range(30):
    sim.step_device(acts[k % 8].data_ptr(), stream=stream)
torch.cuda.synchronize()
assert fn(buf.data_ptr()) == 0
sim.step_device(acts[0].data_ptr(), stream=stream)
torch.cuda.synchronize()
assert fn(None) == 0
t = buf.cpu().numpy().astype(np.float64)
names = ["init", "b0 wait", "phase 1", "b1 wait", "tasks / edge rays", "b2 wait / sweep 0", "sweep 0..", "b3 wait", "sweep 1", "b4 wait", "outputs"]
n_st = 10
d = t[:, :, 1:n_st] - t[:, :, 0:n_st - 1]
print("mean ticks between stamps k-1 -> k, per wave (rows: stamp k = 1..9; columns: wave 0..3)")
for k in range(n_st - 1):
    print(k + 1, np.round(d[:, :, k].mean(axis=0), 1))
print("whole kernel per wave:", np.round((t[:, :, n_st - 1] - t[:, :, 0]).mean(axis=0), 1))
print("pairs: circle", t[:, 0, 14].mean(), "edge", t[:, 0, 15].mean())
