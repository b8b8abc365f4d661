"""Does splitting a batch over two streams pay?  One handle of V envs against two
handles of V/2 envs stepped on two HIP streams (k_vessel of one half can overlap
k_sense of the other).  Prints ms per step of V envs for both."""
import sys
import time

import torch

from gym_auv_amd import _lib, scenarios
from gym_auv_amd.sim import AuvSim


def make(n, rank, stream):
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=n, rank=rank)
    sim = AuvSim(cfg)
    sim.set_option(_lib.OPT_AUTO_RESET, 1)
    sim.set_scene(scene)
    sim.reset(scene["init_state"], stream=stream)
    acts = torch.from_numpy(scenarios.random_actions(n, 8, seed=100 + rank)).cuda()
    return sim, scene, acts


def run(sims, streams, steps):
    for (sim, scene, acts), st in zip(sims, streams):
        sim.reset(scene["init_state"], stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        for (sim, scene, acts), st in zip(sims, streams):
            sim.step_device(acts[k % 8].data_ptr(), stream=st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    V = 65536
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 220
    s0 = torch.cuda.Stream()
    s1 = torch.cuda.Stream()
    one = [make(V, 0, s0)]
    two = [make(V // 2, 0, s0), make(V // 2, 1, s1)]
    four_streams = [torch.cuda.Stream() for _ in range(4)]
    four = [make(V // 4, r, four_streams[r]) for r in range(4)]
    for rep in range(3):
        print("one handle : %.4f ms/step" % run(one, [s0], steps))
        print("two halves : %.4f ms/step" % run(two, [s0, s1], steps))
        print("four parts : %.4f ms/step" % run(four, four_streams, steps))


if __name__ == "__main__":
    main()
