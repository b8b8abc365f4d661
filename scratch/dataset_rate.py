"""config 5: regenerate the LiDAR / risk / metadata files from the dense stress scene; rows per second
through the host logger (page-locked reads, native formatter on the host threads) and through the device
logger (rows compacted and formatted on the GPU): PYTHONPATH=. python scratch/dataset_rate.py [dir]"""
import os, shutil, sys, tempfile, time
import numpy as np
import torch
from gym_auv_amd import scenarios, dataset as D
from gym_auv_amd.sim import AuvSim
cfg, scene = scenarios.make_workload("cfg5_dense_720")
V = cfg["n_envs"]
sim = AuvSim(cfg); sim.set_scene(scene)
st = torch.cuda.current_stream()
acts = torch.from_numpy(scenarios.random_actions(V, 8, seed=3)).cuda()
base = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
for name, cls in (("host logger", D.DatasetLogger), ("device logger", D.DeviceDatasetLogger), ("host logger", D.DatasetLogger),
                  ("device logger", D.DeviceDatasetLogger)):
    sim.reset(scene["init_state"], stream=st)
    d = tempfile.mkdtemp(dir=base)
    w = D.DatasetWriter(d)
    lg = cls(sim, w, every=5)
    for k in range(5):
        sim.step_device(acts[k % 8].data_ptr(), stream=st); lg.after_step(st)
    t0 = time.perf_counter(); r0 = w.rows
    for k in range(25):
        sim.step_device(acts[k % 8].data_ptr(), stream=st); lg.after_step(st)
    dt = time.perf_counter() - t0
    w.close()
    size = sum(os.path.getsize(p) for p in w.paths.values())
    print("cfg5, %s: %d rows (%d rays, <=%d obstacles each) in %.3f s = %.0f rows/s; files %.0f MB total" % (
        name, w.rows - r0, cfg["n_rays"], cfg["n_circles"] + cfg["n_polys"], dt, (w.rows - r0) / dt, size / 1e6), flush=True)
    shutil.rmtree(d)
sim.close()
