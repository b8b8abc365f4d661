"""Time of auv_episode_stats (two launches) at cfg3's size: AUVSIM_LIB=... python scratch/stats_time.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_auv_amd import scenarios
from gym_auv_amd.sim import AuvSim
for name in ("cfg3_65536_moving", "cfg5_dense_720"):
    cfg, scene = scenarios.make_workload(name)
    sim = AuvSim(cfg); sim.set_scene(scene); sim.reset(scene["init_state"])
    out = torch.zeros(8, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream()
    for _ in range(20): sim.episode_stats(out, stream=st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(200): sim.episode_stats(out, stream=st)
    e1.record(st); e1.synchronize()
    print(name, "auv_episode_stats: %.1f us per call" % (e0.elapsed_time(e1) / 200 * 1e3), out.tolist()[:2])
    sim.close()
