"""Where the waves of k_sense spend their time, any workload: mean shader-clock ticks between the phase stamps of a
diagnostic build (-DAUV_PHASE_STAMPS), per wave of the workgroup.
Usage: AUVSIM_LIB=scratch/libauvsim_stamps.so python scratch/stamps2.py WORKLOAD [block [tile]]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_auv_amd import _lib, scenarios
from gym_auv_amd.sim import AuvSim

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5_dense_720"
cfg, scene = scenarios.make_workload(name)
sim = AuvSim(cfg, device=0)
if len(sys.argv) > 2 and int(sys.argv[2]):
    sim.set_option(_lib.OPT_SENSE_BLOCK, int(sys.argv[2]))
if len(sys.argv) > 3:
    sim.set_option(_lib.OPT_RASTER_TILE, int(sys.argv[3]))
sim.set_option(_lib.OPT_AUTO_RESET, 1)
sim.set_scene(scene)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev)
sim.reset(scene["init_state"], stream=stream)
V = cfg["n_envs"]
acts = torch.from_numpy(scenarios.random_actions(V, 8, seed=100)).to(dev)
lib = _lib.load()
geo = (ctypes.c_int32 * 4)()
lib.auv_debug_geometry.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert lib.auv_debug_geometry(sim._h, geo) == 0
te, block, lds = geo[0], geo[1], geo[2]
blocks = (V + te - 1) // te
nw = block // 64
print(f"{name}: tile {te} envs, {block} threads, LDS {lds} B, {blocks} workgroups")
buf = torch.zeros((blocks, nw, 16), dtype=torch.int64, device=dev)
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
names = ["0>1 clear accumulators", "1>2 phase 1 (geometry)", "2>3 wait at barrier", "3>4 phase-2 tasks (task waves)",
         "4>5 sweep 0 (ray waves)", "5>6 wait at barrier", "6>7 sweep 1 + ranges out", "7>8 wait at barrier", "8>9 hit ids out + network"]
d = t[:, :, 1:10] - t[:, :, 0:9]
np.set_printoptions(linewidth=250)
for k in range(9):
    print(f"{names[k]:34s}", np.round(d[:, :, k].mean(axis=0)).astype(int))
print(f"{'whole kernel per wave':34s}", np.round((t[:, :, 9] - t[:, :, 0]).mean(axis=0)).astype(int))
span = t[:, :, 9].max(axis=1) - t[:, :, 0].min(axis=1)
print("workgroup span percentiles 50/90/99/max:", np.percentile(span, [50, 90, 99, 100]).round())
slow = np.argsort(span)[-max(1, len(span) // 100):]
print("the slowest 1 % of the workgroups, mean ticks per phase and wave:")
for k in range(9):
    print(f"{names[k]:34s}", np.round(d[slow][:, :, k].mean(axis=0)).astype(int))
