"""How full the device is over one launch of k_sense: from the phase stamps of a diagnostic build, the number of
workgroups in flight over time (start of the launch, steady state, tail).
Usage: AUVSIM_LIB=scratch/libauvsim_stamps.so python scratch/timeline.py [WORKLOAD]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_auv_amd import _lib, scenarios
from gym_auv_amd.sim import AuvSim

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_65536_moving"
cfg, scene = scenarios.make_workload(name)
sim = AuvSim(cfg, device=0)
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
te, block = geo[0], geo[1]
blocks, nw = (V + te - 1) // te, block // 64
buf = torch.zeros((blocks, nw, 16), dtype=torch.int64, device=dev)
fn = lib.auv_debug_set_stamps
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
for k in range(300):
    sim.step_device(acts[k % 8].data_ptr(), stream=stream)
torch.cuda.synchronize()
assert fn(buf.data_ptr()) == 0
sim.step_device(acts[0].data_ptr(), stream=stream)
torch.cuda.synchronize()
assert fn(None) == 0
t = buf.cpu().numpy().astype(np.float64)
# slots 14 / 15: the device-wide 100 MHz counter at the first and the last stamp of each wave (the shader clocks of
# different CUs are not aligned with one another); in units of 10 ns
start, end = t[:, :, 14].min(axis=1), t[:, :, 15].max(axis=1)
t0, t1 = start.min(), end.max()
span = t1 - t0
print(f"{name}: {blocks} workgroups of {nw} waves; launch span {span / 100:.1f} us; workgroup duration median {np.median(end - start) / 100:.2f} us")
edges = np.linspace(t0, t1, 41)
mid = 0.5 * (edges[1:] + edges[:-1])
active = [(int(((start <= m) & (end > m)).sum())) for m in mid]
print("workgroups in flight at 40 instants across the launch:")
print(active)
area = (end - start).sum()
print(f"mean workgroups in flight {area / span:.0f} (slots: {256 * 8} when 8 fit a CU)")
order = np.argsort(start)
print("start time of workgroup by launch order, every 1024th (x 10 ns from the first):", (start[order][::1024] - t0).round().astype(int).tolist())
late = np.sort(end)[-2048:]
print(f"the last 2048 workgroups end between {(late[0] - t0) / 100:.1f} and {(late[-1] - t0) / 100:.1f} us")
first = np.sort(start)[:2048]
print(f"the first 2048 workgroups start within {(first[-1] - t0) / 100:.2f} us; the first 2048 to end do so at {(np.sort(end)[2047] - t0) / 100:.1f} us")
