"""Can a rollout be captured into a HIP graph as it is?  K steps with resident actions captured on a side stream
(torch.cuda.graph), replayed, compared bit for bit with plain stepping, and timed: PYTHONPATH=. python scratch/graph_steps.py"""
import sys, time
import numpy as np
import torch
from gym_auv_amd import _lib, scenarios
from gym_auv_amd.sim import AuvSim

K = 50
for name in ("cfg1_single_episode", "cfg2_4096_static", "cfg3_65536_moving"):
    cfg, scene = scenarios.make_workload(name)
    V = cfg["n_envs"]
    acts = torch.from_numpy(scenarios.random_actions(V, 8, seed=5)).cuda()
    ptrs = [acts[i].data_ptr() for i in range(8)]
    out = {}
    for mode in ("plain", "graph"):
        sim = AuvSim(cfg); sim.set_scene(scene)
        sim.set_option(_lib.OPT_AUTO_RESET, 1)
        st = torch.cuda.current_stream()
        sim.reset(scene["init_state"], stream=st)
        torch.cuda.synchronize()
        if mode == "plain":
            def rollout():
                s = torch.cuda.current_stream()
                for k in range(K):
                    sim.step_device(ptrs[k % 8], stream=s)
        else:
            g = torch.cuda.CUDAGraph()
            # warm-up launches on a side stream, as torch asks before a capture
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                sim.step_device(ptrs[0], stream=side)
            torch.cuda.synchronize()
            sim.reset(scene["init_state"], stream=st)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                s = torch.cuda.current_stream()
                for k in range(K):
                    sim.step_device(ptrs[k % 8], stream=s)
            rollout = g.replay
            # the capture itself executed nothing: state is still the reset state
        rollout()
        torch.cuda.synchronize()
        out[mode] = {k: sim.read(k).copy() for k in ("state", "ranges", "hit", "risk", "reward", "done", "t_step")}
        t0 = time.perf_counter()
        for _ in range(10):
            rollout()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (10 * K)
        out[mode + "_us"] = dt * 1e6
        sim.close()
    same = all(np.array_equal(out["plain"][k], out["graph"][k], equal_nan=True) for k in out["plain"])
    print(f"{name}: graph replay equals plain stepping: {same}; per step plain {out['plain_us']:.1f} us, graph {out['graph_us']:.1f} us", flush=True)
