"""PCIe-inclusive stepping: actions handed over as host memory every step (auv_step_host)."""
import time
import numpy as np
import torch
from gym_auv_amd import scenarios, cnn
from gym_auv_amd.sim import AuvSim
cfg, scene = scenarios.make_workload("cfg3_65536_moving")
V = cfg["n_envs"]
sim = AuvSim(cfg); sim.set_scene(scene)
st = torch.cuda.current_stream()
acts = scenarios.random_actions(V, 8, seed=100)
for rep in range(2):
    sim.reset(scene["init_state"], stream=st)
    for k in range(20): sim.step_host(acts[k % 8], stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200): sim.step_host(acts[k % 8], stream=st)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("auv_step_host: %.4f ms/step, %.3g env-steps/s" % (dt / 200 * 1e3, V * 200 / dt))
# plus reading the observation back to the host every step
per = np.empty((V, 1, 180), np.float32)
t0 = time.perf_counter()
for k in range(50):
    sim.step_host(acts[k % 8], stream=st)
    per = sim.read("perception", st)
dt = time.perf_counter() - t0
print("auv_step_host + perception to host: %.4f ms/step, %.3g env-steps/s" % (dt / 50 * 1e3, V * 50 / dt))
sim.close()
