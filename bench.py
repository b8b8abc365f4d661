"""Throughput of the simulation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One step = one pass of the whole path (obstacle advance + RKF45 dynamics +
encounter bookkeeping + LiDAR + observation assembly) over one batch of
synthetic envs, actions already resident in HBM.  The default workload is
cfg3: 65,536 vessels per GPU, 8 circles + 8 moving 5-edge ships per env, 180
rays.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every
rank steps its own 65,536 envs (weak scaling, no data-path collective); the
episode-statistic vector is all-reduced once per rollout inside the timed
region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X spec: half the 157.3 TF fp32 vector rate
HBM_PEAK_GBS = 8000.0
FLOP_CIRCLE, FLOP_SEGMENT = 16, 18   # per ray-primitive test (add/mul/cmp/div/sqrt = 1)
DYN_BYTES_PER_VESSEL = 112           # 48 B state in + 48 B out + 16 B action


def cpu_baseline(name, n_envs, steps):
    """The numpy oracle on a bounded sample of the same workload, one core."""
    from gym_auv_amd import scenarios
    from tests.helpers import make_oracle

    cfg, scene = scenarios.make_workload(name, n_envs=n_envs)
    env = make_oracle(cfg, scene)
    env.reset(scene["init_state"])
    acts = scenarios.random_actions(n_envs, steps, seed=7)
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k])
    dt = time.perf_counter() - t0
    return {"value": n_envs * steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n_envs} envs x {steps} steps of {name} (numpy fp64 oracle; substitute for "
                      f"the absent reference env), {dt:.1f} s",
            "rays_per_s": n_envs * steps * cfg["n_rays"] / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3_65536_moving")
    ap.add_argument("--envs", type=int, default=None, help="override vessels per GPU")
    ap.add_argument("--lidar-mode", type=int, default=0, help="0 rasterising (default), 1 all pairs")
    ap.add_argument("--tile", type=int, default=0, help="envs per LiDAR workgroup, 0 = auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=4096)
    ap.add_argument("--cpu-steps", type=int, default=60)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from gym_auv_amd import _lib, scenarios
    from gym_auv_amd.sim import AuvSim

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg, scene = scenarios.make_workload(args.workload, n_envs=args.envs, rank=rank)
    V, R, C, P, NV = cfg["n_envs"], cfg["n_rays"], cfg["n_circles"], cfg["n_polys"], cfg["n_verts"]
    sim = AuvSim(cfg, device=local_rank)
    sim.set_option(_lib.OPT_LIDAR_MODE, args.lidar_mode)
    if args.tile:
        sim.set_option(_lib.OPT_RASTER_TILE, args.tile)
    sim.set_scene(scene)
    stream = torch.cuda.current_stream(dev)
    sim.reset(scene["init_state"], stream=stream)
    # a ring of resident action tensors, [2,V] f64 each
    n_act = 8
    acts = torch.from_numpy(scenarios.random_actions(V, n_act, seed=100 + rank)).to(dev)
    ptrs = [acts[i].data_ptr() for i in range(n_act)]
    done_t, coll_t, rew_t = sim.tensor("done"), sim.tensor("collision"), sim.tensor("reward")

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        sim.step_device(ptrs[k % n_act], stream=stream)
    sync_all()
    # an un-instrumented rehearsal of the same K steps: tells how much the
    # per-kernel HIP events cost (reported as ms_per_step_no_events)
    t0 = time.perf_counter()
    for k in range(args.steps):
        sim.step_device(ptrs[k % n_act], stream=stream)
    sync_all()
    plain_elapsed = time.perf_counter() - t0
    sim.profile(True)
    sim.profile_read()
    t0 = time.perf_counter()
    for k in range(args.steps):
        sim.step_device(ptrs[k % n_act], stream=stream)
    # episode statistics of the rollout: one small all-reduce, not per step
    stats = torch.stack([done_t.sum(dtype=torch.float64), coll_t.sum(dtype=torch.float64),
                         rew_t.sum(), torch.tensor(float(V), device=dev, dtype=torch.float64)])
    if world > 1:
        dist.all_reduce(stats)
    sync_all()
    elapsed = time.perf_counter() - t0
    prof_ms, prof_steps = sim.profile_read()
    sim.profile(False)

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        S = P * NV
        env_steps = V * world * args.steps
        lidar_s = prof_ms["lidar"] / 1e3 / max(prof_steps, 1)
        dyn_s = prof_ms["dynamics"] / 1e3 / max(prof_steps, 1)
        enc_s = prof_ms["encounter"] / 1e3 / max(prof_steps, 1)
        flops = float(FLOP_CIRCLE * C + FLOP_SEGMENT * S) * V * R
        achieved = flops / lidar_s / 1e12
        out = {
            "metric": "vessel_env_steps_per_sec", "value": env_steps / elapsed, "unit": "env-steps/s",
            "rays_per_sec": env_steps * R / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_no_events": plain_elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "vessels_per_gpu": V, "rays": R, "circles": C,
                       "polygons": P, "edges_per_polygon": NV, "sensor_range_m": cfg["sensor_range"],
                       "dt_s": cfg["dt"], "sharding": f"envs x{world}, no data-path collective"},
            "roofline": {"kernel": "k_lidar", "bound": "valu", "achieved": achieved,
                         "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
                         "flops_per_launch": flops, "avg_launch_ms": lidar_s * 1e3,
                         "note": "fp64 vector (non-MFMA) roofline; algorithmic flops = "
                                 "(16*circles + 18*edges) per ray"},
            "roofline_dynamics": {"kernel": "k_dynamics", "bound": "hbm",
                                  "achieved": DYN_BYTES_PER_VESSEL * V / dyn_s / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": DYN_BYTES_PER_VESSEL * V / dyn_s / 1e9 / HBM_PEAK_GBS,
                                  "traffic": None, "avg_launch_ms": dyn_s * 1e3},
            "kernel_ms": {"dynamics": dyn_s * 1e3, "encounter": enc_s * 1e3, "lidar": lidar_s * 1e3},
            "episode_stats": {"done": stats[0].item(), "collisions": stats[1].item(),
                              "reward_sum": stats[2].item(), "envs": stats[3].item()},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_envs, args.cpu_steps)
        print(json.dumps(out), flush=True)
    sim.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
