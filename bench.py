"""Throughput of the simulation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One step = one pass of the whole path (RKF45 dynamics + navigation, obstacle
advance, distances / collision / metadata / risk, LiDAR, observation assembly)
over one batch of synthetic envs, actions already resident in HBM.  The default
workload is cfg3: 65,536 vessels per GPU, 8 circles + 8 moving 5-edge ships per
env, 180 rays.  With N > 1 every rank steps its own 65,536 envs (weak scaling,
no data-path collective); the episode-statistic vector is all-reduced once per
rollout inside the timed region, and after the timed rollouts every rank's last
scans are gathered on the trainer rank 0 (timed on its own and reported as
`scan_gather`, not part of `value`).  Rank 0 prints ONE JSON line.  `--gpus N` run
as a plain `python bench.py` starts its N ranks itself (torch.distributed.run,
one rank per GPU, rendezvous on 127.0.0.1) before anything touches a GPU; run
under torch.distributed.run it takes the ranks it is given, and refuses a
WORLD_SIZE that differs from N.

The timed rollout (W warm-up steps from the reset state, then K timed steps) is
repeated `--repeats` times; `value` and `ms_per_step` are the median repeat,
every repeat being the maximum over ranks.

The timed region carries no instrumentation.  The per-kernel durations behind
`roofline` come from a second pass of the same steps (max(K, 200) of them), in
the same process, with HIP events recorded around every kernel on the launch
stream (that pass is slower by the cost of the events and is reported as
ms_per_step_with_events).
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
# device memory is shared between the ranks' processes through dmabuf handles (what this pool's driver supports);
# set before anything loads the HIP runtime, inherited by the ranks this script starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X spec: half the 157.3 TF fp32 vector rate
HBM_PEAK_GBS = 8000.0
FLOP_CIRCLE, FLOP_SEGMENT = 16, 18   # per ray-primitive test (add/mul/cmp/div/sqrt = 1)


def sense_bytes_per_env(R, C, P, NV, meta_fields=5):
    """Algorithmic HBM bytes of one k_sense launch per env-step: scene and
    vessel record in, advanced positions back, every output out (no re-reads).
    meta_fields: columns of a metadata record written per step (with the
    reference's index the heading column is a constant of the scene and is
    not rewritten: 4)."""
    O = C + P
    scene_in = C * 40 + P * (48 + NV * 16 + 4 + 8 + 8)
    vessel_in = 48 + 64 + 2
    pos_out = O * 16
    lidar_out = R * 16
    encounter_out = O * (8 + 8 + 8 * meta_fields) + (1 + 1 + 8 + 4 + 8)
    return scene_in + vessel_in + pos_out + lidar_out + encounter_out


VESSEL_BYTES_PER_ENV = 48 + 48 + 16 + 48 + 64 + 16 + 8 + 8 + 24 + 2 + 8  # state rw, action, path, record, own, along rw, nav, flags, t_step


def cpu_baseline(name, n_envs, steps, threads, scalar_envs=4):
    """The oracle on a bounded sample of the same workload, on the host cores,
    in three forms: (`value`) the C restatement (oracle/auv_oracle.c, scalar
    per env, all ray-primitive pairs) with `threads` threads over disjoint env
    ranges; the numpy oracle vectorised over envs on one core; and the numpy
    oracle as one object per env stepped one after the other on one core,
    which is how a gym env runs."""
    from gym_auv_amd import params, scenarios
    from oracle import c_oracle
    from tests.helpers import make_oracle

    from gym_auv_amd.scenarios import WORKLOADS
    wl = WORKLOADS[name]
    per_env = wl["R"] * (wl["C"] + wl["P"] * wl["NV"])  # ray-primitive pairs of one env
    # every leg is sized by the scene: the pairs of `steps` steps of the default scene (8,640 per env) at most
    budget = n_envs * steps * 8640 // per_env  # env-steps
    steps = max(5, min(steps, budget // n_envs))
    n_envs = max(16 * threads, min(n_envs, budget // steps))
    print(f"[bench] cpu baseline: C oracle, {n_envs} envs x {steps} steps, {threads} threads", file=sys.stderr, flush=True)
    cfg, scene = scenarios.make_workload(name, n_envs=n_envs)
    env = c_oracle.COracleEnv(cfg, params.ray_table(cfg["n_rays"]), threads=threads)
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"],
                  scene.get("poly_heading"))
    env.set_path(scene["path"])
    env.reset(scene["init_state"])
    acts = scenarios.random_actions(n_envs, steps, seed=7)
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k])
    dt = time.perf_counter() - t0
    env.close()
    # the numpy legs are sized by the scene: about as many ray-primitive pairs as 2,048 envs of the default scene
    n_np, s_np = max(16, min(n_envs, 2048, int(2048 * 8640 / per_env))), min(steps, 10)
    print(f"[bench] cpu baseline: numpy oracle, {n_np} envs x {s_np} steps, then one object per env", file=sys.stderr,
          flush=True)
    cfg2, scene2 = scenarios.make_workload(name, n_envs=n_np)
    env2 = make_oracle(cfg2, scene2)
    env2.reset(scene2["init_state"])
    t1 = time.perf_counter()
    for k in range(s_np):
        env2.step(acts[k][:, :n_np])
    dt_np = time.perf_counter() - t1
    # one env per object, as a gym env is stepped
    s_sc = min(steps, 25)
    singles = []
    for e in range(scalar_envs):
        one = {k: (v[:, e:e + 1] if k == "init_state" else v[e:e + 1]) for k, v in scene2.items()}
        cfg1 = dict(cfg2, n_envs=1)
        env1 = make_oracle(cfg1, one)
        env1.reset(one["init_state"])
        singles.append(env1)
    t2 = time.perf_counter()
    for e, env1 in enumerate(singles):
        for k in range(s_sc):
            env1.step(acts[k][:, e:e + 1])
    dt_sc = time.perf_counter() - t2
    return {"value": n_envs * steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n_envs} envs x {steps} steps of {name}, C oracle (oracle/auv_oracle.c), {threads} threads, "
                      f"{dt:.1f} s; substitute for the absent reference env",
            "rays_per_s": n_envs * steps * cfg["n_rays"] / dt,
            "numpy_one_core": {"value": n_np * s_np / dt_np, "unit": "env-steps/s", "cores": 1,
                               "sample": f"{n_np} envs x {s_np} steps, vectorised over envs, {dt_np:.1f} s"},
            "numpy_scalar_per_env": {"value": scalar_envs * s_sc / dt_sc, "unit": "env-steps/s", "cores": 1,
                                     "sample": f"{scalar_envs} envs x {s_sc} steps, one oracle object per env "
                                               f"stepped in turn, {dt_sc:.1f} s"}}


def measured_traffic(key, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary for this
    exact workload, or None (the counters are collected by separate rocprofv3
    passes, not inside this run)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f)[key][kernel]
        return (2.0 * e["fetch_kib"] + e["write_kib"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def measured_valu_insts(key, kernel):
    """VALU wave-instructions per launch from the committed PMC summary, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return float(json.load(f)[key][kernel]["valu_insts"])
    except (OSError, KeyError, ValueError):
        return None


def measured_counter_ratios(key, kernel):
    """Ratios of the committed PMC passes that need no clock or cycles-per-instruction assumption: the share of the
    device's VALU issue cycles the kernel used, the share of wave cycles spent waiting, LDS bank conflicts per
    LDS-active cycle; {} when the workload has no committed pass."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f)[key][kernel]
        return {k: e[k] for k in ("valu_busy", "wait_frac", "lds_conflict_frac") if k in e}
    except (OSError, KeyError, ValueError):
        return {}


# one VALU instruction per SIMD every 4 cycles (a 64-wide wave on 16 lanes), 4 SIMDs x 256 CUs, 2.4 GHz
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0


def launch_ranks(n):
    """Re-run this script as n ranks under torch.distributed.run and return its
    exit status.  Nothing has imported torch or touched a GPU at this point."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def rehearse_without_gpu(args, world, rank):
    """AUV_BENCH_REHEARSE=nogpu: the launch, rendezvous, barrier, statistics all-reduce and max-over-ranks
    timing of a multi-rank run on a machine without a GPU (gloo); no simulation runs and the line says so."""
    import torch
    import torch.distributed as dist

    from gym_auv_amd import dist as auv_dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    vec = torch.zeros(len(auv_dist.STAT_FIELDS), dtype=torch.float64)
    vec[0] = 1.0
    totals = auv_dist.reduce_episode_stats(vec)
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # the scan gather of the multi-rank line, on stand-in scans (64 envs per rank, each rank's filled with its number)
    mine = torch.full((64, 1, 180), float(rank + 1))
    gathered = torch.empty((64 * world, 1, 180)) if rank == 0 else None
    t1 = time.perf_counter()
    auv_dist.gather_scans(mine, dst=0, out=gathered)
    g_ms = (time.perf_counter() - t1) * 1e3
    ok = None if rank else bool(torch.equal(gathered[:, 0, 0].reshape(world, 64)[:, 0], torch.arange(1.0, world + 1.0)))
    if rank == 0:
        print(json.dumps({"metric": "vessel_env_steps_per_sec", "value": 0.0, "unit": "env-steps/s", "n_gpus": world,
                          "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f64", "data": "rehearsal without a GPU: no simulation ran",
                          "config": {"workload": args.workload or "cfg4_8x65536_moving",
                                     "sharding": f"envs x{world}, no data-path collective"},
                          "ranks_joined": int(totals["env_steps"]),
                          "collectives": {"backend": "gloo", "stats_allreduce_ms": float(t[0]) * 1e3,
                                          "scan_gather": {"ms": g_ms, "bytes_inbound": 64.0 * 180 * 4 * (world - 1),
                                                          "checked": ok}}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if int(totals["env_steps"]) == args.gpus else 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed rollouts; the median one is reported")
    ap.add_argument("--workload", default=None,
                    help="default: cfg3_65536_moving on one GPU, cfg4_8x65536_moving (the same scene family, "
                         "65,536 vessels on every GPU) on several")
    ap.add_argument("--envs", type=int, default=None, help="override vessels per GPU")
    ap.add_argument("--lidar-mode", type=int, default=0, help="0 rasterising (default), 1 all pairs")
    ap.add_argument("--tile", type=int, default=0, help="envs per k_sense workgroup, 0 = auto")
    ap.add_argument("--block", type=int, default=0, help="threads per k_sense workgroup (64/128/256), 0 = default")
    ap.add_argument("--cnn", default="shallow", choices=["none", "shallow", "deep2", "deep"],
                    help="LiDAR CNN fed zero-copy with every step's scans (features head)")
    ap.add_argument("--cnn-mode", default="attached", choices=["attached", "separate"],
                    help="attached: evaluated inside the sensing kernel; separate: its own launch per step")
    ap.add_argument("--spin-seconds", type=float, default=5.0,
                    help="upper bound of the untimed stepping that brings a cold GPU to its working clocks before the "
                         "warm-up: blocks of 50 steps until two consecutive blocks agree within 2 %% "
                         "(0 under a profiler, so that its per-kernel averages cover the measured rollout only)")
    ap.add_argument("--no-auto-reset", action="store_true",
                    help="leave finished episodes (collided / arrived / truncated envs) as they are instead of "
                         "restarting them inside the next step, as a vectorised RL env does")
    ap.add_argument("--meta-every", type=int, default=1,
                    help="refresh the per-obstacle bearing/course/risk every n-th step only (default 1: every step)")
    ap.add_argument("--risk-model", default="reference", choices=["reference", "cpa"],
                    help="risk index: the reference's own (recovered from its dataset files; default), or the "
                         "self-defined closest-point-of-approach index")
    ap.add_argument("--dataset-rows", default="auto", choices=["auto", "on", "off"],
                    help="after the timed rollouts, regenerate dataset rows (LiDAR / risk / metadata text files) "
                         "from the running batch for 25 steps and report rows/s; auto: for the dense cfg5 workload, "
                         "whose purpose that is")
    ap.add_argument("--gather-scans", default="auto", choices=["auto", "on", "off"],
                    help="gather the last scans of every rank on the trainer rank 0 after the timed rollouts and "
                         "report its time and rate (auto: whenever there is a process group)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU baseline (0 = the CPUs this process may use, at most 16)")
    ap.add_argument("--cpu-envs", type=int, default=32768)
    ap.add_argument("--cpu-steps", type=int, default=200)
    ap.add_argument("--cpu-scalar-envs", type=int, default=4,
                    help="envs of the scalar-per-env leg of the CPU baseline (numpy oracle, one env per object)")
    args = ap.parse_args()
    if args.gpus < 1 or args.repeats < 1:
        ap.error("--gpus and --repeats must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"error: --gpus {args.gpus} but WORLD_SIZE is {world}: start {args.gpus} ranks "
                  f"(python bench.py --gpus {args.gpus} does it by itself)", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("AUV_BENCH_WATCHDOG"):
        # diagnostic: dump every thread's stack and exit if the run is still going after that many seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["AUV_BENCH_WATCHDOG"]), exit=True)
    if os.environ.get("AUV_BENCH_REHEARSE") == "nogpu":
        sys.exit(rehearse_without_gpu(args, world, rank))

    import torch
    import torch.distributed as dist

    from gym_auv_amd import _lib, scenarios
    from gym_auv_amd import dist as auv_dist
    from gym_auv_amd.sim import AuvSim

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank path on a one-GPU box: AUV_BENCH_REHEARSE=1 puts
    # every rank on cuda:0 and uses gloo (RCCL refuses two ranks on one device)
    rehearse = os.environ.get("AUV_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    # AUV_BENCH_FORCE_GROUP=1: create the process group on a single rank too, so that a one-GPU box runs the very
    # collectives (RCCL initialisation, barrier, statistics all-reduce, scan gather) of the multi-GPU run
    grouped = world > 1 or os.environ.get("AUV_BENCH_FORCE_GROUP") == "1"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if grouped and dist.get_world_size() != args.gpus:
        sys.exit(3)   # fewer ranks joined than asked for
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.workload is None:
        args.workload = "cfg3_65536_moving" if world == 1 else "cfg4_8x65536_moving"
    cfg, scene = scenarios.make_workload(args.workload, n_envs=args.envs, rank=rank)
    V, R, C, P, NV = cfg["n_envs"], cfg["n_rays"], cfg["n_circles"], cfg["n_polys"], cfg["n_verts"]
    cfg["risk_model"] = _lib.RISK_REFERENCE if args.risk_model == "reference" else _lib.RISK_CPA
    sim = AuvSim(cfg, device=local_rank)
    sim.set_option(_lib.OPT_LIDAR_MODE, args.lidar_mode)
    if args.tile:
        sim.set_option(_lib.OPT_RASTER_TILE, args.tile)
    if args.block:
        sim.set_option(_lib.OPT_SENSE_BLOCK, args.block)
    auto_reset = not args.no_auto_reset and args.lidar_mode == 0
    if auto_reset:
        sim.set_option(_lib.OPT_AUTO_RESET, 1)
    if args.meta_every != 1:
        sim.set_option(_lib.OPT_META_EVERY, args.meta_every)
    sim.set_scene(scene)
    stream = torch.cuda.current_stream(dev)
    sim.reset(scene["init_state"], stream=stream)
    # a ring of resident action tensors, [2,V] f64 each
    n_act = 8
    acts = torch.from_numpy(scenarios.random_actions(V, n_act, seed=100 + rank)).to(dev)
    ptrs = [acts[i].data_ptr() for i in range(n_act)]
    per_all = sim.tensor("perception")

    # the consumer of the scans: the reference's shipped weights (fixture made from its own files)
    net = feats = per_t = None
    cnn_dim = 0   # outputs per scan of a network attached to the sensing kernel
    if args.cnn != "none" and R == 180:
        import numpy as np

        from gym_auv_amd import cnn

        gold = np.load(os.path.join(ROOT, "tests", "golden", "cnn_golden.npz"))
        pre = f"w_{args.cnn}/"
        net = cnn.LidarCNN(args.cnn, {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)},
                           device=local_rank)
        if args.cnn_mode == "attached":
            sim.attach_cnn(net, "features")
            cnn_dim = net.out_dim("features")
            net.close()
            net = None
        else:
            per_t = sim.tensor("perception")
            feats = torch.empty((V, net.out_dim("features")), dtype=torch.float32, device=dev)

    def sync_all():
        torch.cuda.synchronize(dev)
        if grouped:
            dist.barrier()
            torch.cuda.synchronize(dev)

    rollout_stats = auv_dist.EpisodeStats(dev)

    def episode_stats():
        # episode statistics of a rollout: two small launches and one small all-reduce (RCCL), not per step; the
        # totals stay on the device, the bracket's synchronisation covers them and they are read after the timing
        rollout_stats.update_from_sim(sim, stream=stream, accumulate=False)
        return rollout_stats.reduce_on_device()

    def rollout():
        for k in range(args.steps):
            sim.step_device(ptrs[k % n_act], stream=stream)
            if net is not None:
                net.forward(per_t, "features", out=feats)
        return episode_stats()

    # bring a cold GPU to its working clocks first, then the W warm-up steps (a fresh process on an idle
    # box runs its first second about 2x slow and keeps ramping for a while): blocks of 50 steps, each from the
    # reset state so that they do the same work, until two consecutive blocks agree within 2 %
    t_spin = time.perf_counter()
    spin_blocks = []
    while time.perf_counter() - t_spin < args.spin_seconds:
        sim.reset(scene["init_state"], stream=stream)
        torch.cuda.synchronize(dev)
        tb = time.perf_counter()
        for k in range(50):
            sim.step_device(ptrs[k % n_act], stream=stream)
            if net is not None:
                net.forward(per_t, "features", out=feats)
        torch.cuda.synchronize(dev)
        spin_blocks.append((time.perf_counter() - tb) / 50 * 1e3)
        if time.perf_counter() - t_spin >= min(0.3, args.spin_seconds) and len(spin_blocks) >= 4 \
                and abs(spin_blocks[-1] - spin_blocks[-2]) <= 0.02 * spin_blocks[-2] \
                and abs(spin_blocks[-1] - spin_blocks[-3]) <= 0.02 * spin_blocks[-3]:
            break
    spin_s = time.perf_counter() - t_spin
    def prepare():
        # A measured rollout starts from the reset state (whatever came before did to the episodes) and runs the W
        # warm-up steps: every pass sees the same scene.  Before it, untimed, as many steps again as the rollout
        # has (150 at least): the reset of a short rollout would otherwise follow a host-side pause long enough
        # for the clocks to drop.  The reset takes the initial states already on the device, so nothing here
        # waits for the host.
        for k in range(max(150, args.steps) if args.spin_seconds > 0 else 0):
            sim.step_device(ptrs[k % n_act], stream=stream)
        sim.reset(None, stream=stream)
        for k in range(args.warmup):
            sim.step_device(ptrs[k % n_act], stream=stream)

    prepare()
    episode_stats()   # sets up the communicator outside the timed region
    times = []
    for rep in range(args.repeats):
        prepare()
        sync_all()
        t0 = time.perf_counter()
        stats = rollout()
        sync_all()
        times.append(time.perf_counter() - t0)
    stats = auv_dist.totals_as_dict(stats)

    # the two collectives on their own: the statistics all-reduce (which the timed region above contains) and the
    # gather of every rank's last scans on the trainer rank 0; barrier-bracketed like the rollouts, 5 repeats each
    collectives = None
    want_gather = args.gather_scans == "on" or (args.gather_scans == "auto" and grouped)
    if grouped or want_gather:
        def timed(fn, reps=5):
            out = []
            for _ in range(reps):
                sync_all()
                t0 = time.perf_counter()
                fn()
                sync_all()
                out.append(time.perf_counter() - t0)
            tt = torch.tensor(out, device=dev, dtype=torch.float64)
            auv_dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return sorted(tt.tolist())[reps // 2] * 1e3

        sync_ms = timed(lambda: None)            # what the bracket itself costs
        collectives = {"stats_allreduce_ms": timed(episode_stats), "bracket_ms": sync_ms}
        if want_gather:
            gathered = torch.empty((V * world, 1, R), dtype=torch.float32, device=dev) if rank == 0 else None
            auv_dist.gather_scans(per_all, dst=0, out=gathered)     # first call: connections set up
            g_ms = timed(lambda: auv_dist.gather_scans(per_all, dst=0, out=gathered))
            # every rank's checksum against the slice the trainer holds
            # (the bit patterns added as integers: exact, whatever the order of the addition)
            sums = torch.zeros(world, dtype=torch.int64, device=dev)
            sums[rank] = per_all.view(torch.int32).sum(dtype=torch.int64)
            auv_dist.all_reduce(sums)
            ok = None
            if rank == 0:
                ok = bool(torch.equal(gathered.view(torch.int32).reshape(world, -1).sum(dim=1, dtype=torch.int64), sums)) \
                    and bool(torch.equal(gathered[:V], per_all))
            inbound = float(V) * R * 4 * (world - 1)
            collectives["scan_gather"] = {
                "ms": g_ms, "bytes_inbound": inbound, "bytes_per_rank": float(V) * R * 4,
                "GB/s": inbound / max(g_ms - sync_ms, 1e-6) / 1e6,
                "frac_of_7x153": inbound / max(g_ms - sync_ms, 1e-6) / 1e6 / (7 * 153.0),
                "checked": ok,
                "what": "every rank's last scans [V,1,R] float32 to the trainer rank 0 (dist.gather_scans: peer-to-peer "
                        "gather into the slices of one tensor), median of 5, barrier-bracketed; the rate is over the "
                        "time less the bracket's own, against the trainer's 7 inbound xGMI links x 153 GB/s"}
            del gathered

    # second pass, instrumented: HIP events around every kernel, on the launch stream; 200 steps at least, so that
    # a short run's per-kernel averages are taken over as many launches as a long run's
    ev_steps = max(args.steps, 200)
    prepare()
    sim.profile(True)
    sim.profile_read()
    t0 = time.perf_counter()
    for k in range(ev_steps):
        sim.step_device(ptrs[k % n_act], stream=stream)
        if net is not None:
            net.forward(per_t, "features", out=feats)
    episode_stats()
    sync_all()
    ev_elapsed = time.perf_counter() - t0
    prof_ms, prof_steps = sim.profile_read()
    sim.profile(False)
    cnn_ms = None
    if net is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(args.steps):
            net.forward(per_t, "features", out=feats)
        e1.record(stream)
        e1.synchronize()
        cnn_ms = e0.elapsed_time(e1) / args.steps

    # what cfg5 is for: regenerating the dataset files from the running batch (outside every timed region above)
    dataset_rate = None
    if rank == 0 and (args.dataset_rows == "on" or (args.dataset_rows == "auto" and args.workload.startswith("cfg5"))):
        import shutil
        import tempfile

        from gym_auv_amd import dataset as D

        def regenerate(logger_cls, passes):
            """rows/s of the last of `passes` identical runs (the first one allocates the staging buffers)"""
            for _ in range(passes):
                tmp = tempfile.mkdtemp(prefix="auv_rows_")
                try:
                    sim.reset(scene["init_state"], stream=stream)
                    w = D.DatasetWriter(tmp)
                    lg = logger_cls(sim, w)
                    t0 = time.perf_counter()
                    for k in range(25):
                        sim.step_device(ptrs[k % n_act], stream=stream)
                        lg.after_step(stream)
                    dt_rows = time.perf_counter() - t0
                    w.close()
                    size = sum(os.path.getsize(pth) for pth in w.paths.values())
                finally:
                    shutil.rmtree(tmp, ignore_errors=True)
            return w.rows, dt_rows, size

        rows_d, dt_d, size = regenerate(D.DeviceDatasetLogger, 2)
        rows_h, dt_h, _ = regenerate(D.DatasetLogger, 1)
        dataset_rate = {"rows_per_s": rows_d / dt_d, "rows": rows_d, "seconds": dt_d, "text_MB": size / 1e6,
                        "what": "25 steps, every env logged every 5th step of its episode through DeviceDatasetLogger "
                                "(obstacle lists, row compaction and the %.18e text on the GPU; the host writes the "
                                f"bytes): {R} ranges + risk + metadata per row",
                        "host_logger_rows_per_s": rows_h / dt_h,
                        "host_logger": "the same rows through DatasetLogger (page-locked reads, native formatter on "
                                       "the host threads); same files byte for byte"}

    t = torch.tensor(times, device=dev, dtype=torch.float64)
    auv_dist.all_reduce(t, op=dist.ReduceOp.MAX)   # every repeat: the slowest rank
    times = sorted(t.tolist())
    elapsed = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])

    if rank == 0:
        S = P * NV
        env_steps = V * world * args.steps
        n = max(prof_steps, 1)
        sense_s = prof_ms["lidar"] / 1e3 / n
        vessel_s = prof_ms["dynamics"] / 1e3 / n
        enc_s = prof_ms["encounter"] / 1e3 / n
        fused = args.lidar_mode == 0
        flops = float(FLOP_CIRCLE * C + FLOP_SEGMENT * S) * V * R
        out = {
            "metric": "vessel_env_steps_per_sec", "value": env_steps / elapsed, "unit": "env-steps/s",
            "rays_per_sec": env_steps * R / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": args.repeats, "ms_per_step_all": [x / args.steps * 1e3 for x in t.tolist()],
            "ms_per_step_with_events": ev_elapsed / ev_steps * 1e3, "steps_with_events": ev_steps,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "vessels_per_gpu": V, "rays": R, "circles": C,
                       "polygons": P, "edges_per_polygon": NV, "sensor_range_m": cfg["sensor_range"],
                       "dt_s": cfg["dt"], "lidar": "rasterising, fused k_sense" if fused else "all pairs, 3 kernels",
                       "cnn": (f"{args.cnn} (features head) on every step's scans, separate launch, zero-copy"
                               if net is not None else
                               f"{args.cnn} (features head) on every step's scans, inside k_sense" if cnn_dim
                               else "none"),
                       "meta_every": args.meta_every, "risk_model": args.risk_model,
                       "auto_reset": "finished episodes restart inside the next step, on the device" if auto_reset
                                     else "off",
                       "sharding": f"envs x{world}, no data-path collective"},
            "sense_launch": sim.launch_info() if fused else None,
            "kernel_ms": ({"k_vessel": vessel_s * 1e3, "k_sense": sense_s * 1e3} if fused else
                          {"k_dynamics": vessel_s * 1e3, "k_encounter": enc_s * 1e3, "k_lidar": sense_s * 1e3}),
            "episode_stats": stats,
            "ranks_joined": dist.get_world_size() if grouped else 1,
            "spin": {"seconds": spin_s, "blocks": len(spin_blocks), "ms_per_step_blocks": spin_blocks[:4] + spin_blocks[-4:],
                     "rule": "blocks of 50 steps from the reset state, for 0.3 s at least and until the last block is "
                             f"within 2 % of each of the two before it, {args.spin_seconds:g} s at most"},
        }
        if collectives is not None:
            out["collectives"] = dict(collectives, backend=dist.get_backend())
        if dataset_rate is not None:
            out["dataset_regeneration"] = dataset_rate
        if cnn_ms is not None:
            out["kernel_ms"]["k_cnn"] = cnn_ms
            cb = float(V) * (720 + 4 * net.out_dim("features"))
            out["roofline_cnn"] = {"kernel": "k_cnn", "bound": "hbm", "achieved": cb / cnn_ms / 1e6,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cb / cnn_ms / 1e6 / HBM_PEAK_GBS,
                                   "traffic": None, "avg_launch_ms": cnn_ms}
        # cfg4 is cfg3's scene family and shape on every rank: the per-launch counters of one GPU apply to it
        twork = "cfg3_65536_moving" if args.workload == "cfg4_8x65536_moving" else args.workload
        tkey = f"{twork}/{V}/" + (f"{args.cnn}-attached" if cnn_dim else
                                  f"{args.cnn}-separate" if net is not None else "none")
        if args.risk_model != "reference":
            tkey += "/" + args.risk_model     # the committed counters are those of the default index
        if fused:
            meta_fields = 4 if (args.risk_model == "reference" and args.meta_every == 1) else 5
            nbytes = float(sense_bytes_per_env(R, C, P, NV, meta_fields) + 4 * cnn_dim) * V
            gbs = nbytes / sense_s / 1e9
            out["roofline"] = {
                "kernel": "k_sense", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": measured_traffic(tkey, "k_sense"),
                "traffic_source": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) from the committed rocprofv3 PMC "
                                  "passes over this workload (profiles/traffic.json), not counted in this run",
                "bytes_per_launch": nbytes,
                "avg_launch_ms": sense_s * 1e3,
                "note": "rasterised LiDAR + encounter + obstacle advance in one pass over the scene; "
                        "algorithmic bytes = scene + vessel record in, positions + all outputs out"}
            vi = measured_valu_insts(tkey, "k_sense")
            if vi is not None:
                out["valu_issue"] = {
                    "kernel": "k_sense", "achieved": vi / sense_s, "peak": VALU_ISSUE_PEAK, "unit": "wave-instr/s",
                    "frac": vi / sense_s / VALU_ISSUE_PEAK, "insts_per_launch": vi,
                    "from_counters": measured_counter_ratios(tkey, "k_sense"),
                    "note": "executed VALU wave-instructions per launch, from the committed PMC pass over a "
                            "rollout of this workload (profiles/traffic.json), over THIS run's average launch "
                            f"time ({ev_steps} steps), against the issue rate of 1024 SIMDs at 2.4 GHz with 4 cycles "
                            "per instruction; from_counters.valu_busy is the same share taken from cycle counters "
                            "alone (SQ_ACTIVE_INST_VALU x 4 over the launch's GRBM_GUI_ACTIVE cycles x 1024 SIMDs): "
                            "the figure to quote"}
            vb = float(VESSEL_BYTES_PER_ENV) * V
            out["roofline_vessel"] = {
                "kernel": "k_vessel", "bound": "hbm", "achieved": vb / vessel_s / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": vb / vessel_s / 1e9 / HBM_PEAK_GBS,
                "traffic": measured_traffic(tkey, "k_vessel"), "avg_launch_ms": vessel_s * 1e3}
            out["allpairs_equivalent"] = {
                "bound": "valu", "achieved": flops / sense_s / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": flops / sense_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "note": "(16*circles + 18*edges) flop per ray, the work an all-pairs ray-cast would do; "
                        "k_sense tests only the rays each primitive can touch, so this is an "
                        "equivalent rate, not executed flops"}
        else:
            ach = flops / sense_s / 1e12
            out["roofline"] = {
                "kernel": "k_lidar", "bound": "valu", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
                "flops_per_launch": flops, "avg_launch_ms": sense_s * 1e3,
                "note": "fp64 vector (non-MFMA) roofline; (16*circles + 18*edges) flop per ray"}
        if not args.no_cpu_baseline and world == 1:
            threads = args.cpu_threads or max(1, min(16, len(os.sched_getaffinity(0))))
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_envs, args.cpu_steps, threads,
                                               args.cpu_scalar_envs)
        print(json.dumps(out), flush=True)
    sim.close()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
