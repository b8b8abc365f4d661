"""Diagnostic: per-phase shader-clock spans of k_sense and pair-list sizes (stamps build)."""
import os, sys, ctypes as C
sys.path.insert(0, '.')
os.environ["AUVSIM_LIB"] = os.path.abspath("scratch/libauvsim_stamps.so")
import torch, numpy as np
from gym_auv_amd import scenarios, _lib
from gym_auv_amd.sim import AuvSim
V = 65536
cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=V)
sim = AuvSim(cfg); sim.set_scene(scene)
st = torch.cuda.current_stream()
sim.reset(scene["init_state"], stream=st)
acts = torch.from_numpy(scenarios.random_actions(V, 8, seed=100)).cuda()
for k in range(100): sim.step_device(acts[k % 8].data_ptr(), stream=st)
nblk = V // 4
stamps = torch.zeros(nblk * 4 * 16, dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.auv_debug_set_stamps.argtypes = [C.c_void_p]
assert lib.auv_debug_set_stamps(C.c_void_p(stamps.data_ptr())) == 0
sim.step_device(acts[0].data_ptr(), stream=st)
torch.cuda.synchronize()
raw = stamps.cpu().numpy().reshape(nblk, 4, 16)
ncp, nep = raw[:, 0, 14], raw[:, 0, 15]
tot = ncp + nep
print("pairs per block: circle mean %.0f edge mean %.0f total mean %.0f p50 %d p90 %d p99 %d max %d; >1024: %.3f" % (
    ncp.mean(), nep.mean(), tot.mean(), *np.percentile(tot, [50, 90, 99]), tot.max(), (tot > 1024).mean()))
ok = (raw[:, :, 5] != 0).all(axis=1)          # blocks whose pair list fitted (the others skip stamps 5..8)
raw = raw[ok]
s = raw[:, :, :10].astype(np.float64)
d = np.diff(s, axis=2)
names = ["P0 init", "P1 stage", "bar1", "P2 obst", "sweep0", "bar2", "sweep1", "bar3", "output"]
for w in range(4):
    print("wave", w, " ".join(f"{n}:{d[:, w, i].mean():7.0f}" for i, n in enumerate(names)))
print("all  ", " ".join(f"{n}:{d[:, :, i].mean():7.0f}" for i, n in enumerate(names)))
print("block span", (s[:, :, 9].max(1) - s[:, :, 0].min(1)).mean())
