"""The N > 1 path on CPU: world_size-2 gloo process group, block sharding of
envs, the rollout statistics all-reduce and the scan gather.  Each rank steps
its own shard with the CPU oracle (no GPU here); the union must equal a
single-process run of the whole batch, bit for bit, because envs never
interact."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from gym_auv_amd import dist as D
from gym_auv_amd import scenarios
from tests.helpers import make_oracle

V, STEPS = 37, 4     # odd on purpose: shards of 19 and 18


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_shard(cfg, scene, acts):
    env = make_oracle(cfg, scene)
    env.reset(scene["init_state"])
    stats = D.EpisodeStats("cpu")
    for k in range(STEPS):
        env.step(acts[k])
        stats.update(torch.from_numpy(env.done), torch.from_numpy(env.collision),
                     torch.from_numpy(env.reached_goal), torch.from_numpy(env.truncated),
                     torch.from_numpy(env.reward), torch.from_numpy(env.ranges), torch.from_numpy(env.risk))
    return env, stats


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=V)
        acts = scenarios.random_actions(V, STEPS, seed=3)
        b, e = D.shard_range(V, rank, world)
        env, stats = _run_shard(dict(cfg, n_envs=e - b), D.shard_scene(scene, rank, world), acts[:, :, b:e])
        totals = stats.reduce()
        # the totals left in a tensor (read back whenever the caller likes) are the same numbers
        assert D.totals_as_dict(stats.reduce_on_device()) == totals
        scans = D.gather_scans(torch.from_numpy(env.perception), dst=0)
        if rank == 0:
            q.put((totals, scans.numpy()))
        else:
            assert scans is None
            q.put((totals, None))
    finally:
        tdist.destroy_process_group()


def test_shard_ranges_partition():
    for n in (1, 7, 64, 65536 * 8 + 3):
        for w in (1, 2, 3, 8):
            r = [D.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_rollout_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=V)
    env, stats = _run_shard(cfg, scene, scenarios.random_actions(V, STEPS, seed=3))
    want = dict(zip(D.STAT_FIELDS, stats.vec.tolist()))
    scans = [s for _, s in got if s is not None]
    assert len(scans) == 1 and np.array_equal(scans[0], env.perception)
    for totals, _ in got:            # every rank holds the node-wide totals
        assert totals["env_steps"] == V * STEPS
        for k in D.STAT_FIELDS:
            assert totals[k] == pytest.approx(want[k], rel=1e-12)


def test_single_process_paths_need_no_group():
    x = torch.rand(5, 1, 180)
    assert D.gather_scans(x) is x
    s = D.EpisodeStats("cpu")
    z = torch.zeros(5, dtype=torch.uint8)
    s.update(z, z, z, z, torch.ones(5, dtype=torch.float64))
    assert s.reduce()["reward_sum"] == 5.0


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = torch.full((6, 1, 180), float(rank + 1)) + torch.arange(180.0) / 1000.0
        out = torch.zeros((12, 1, 180)) if rank == 0 else None
        got = D.gather_scans(mine, dst=0, out=out)
        if rank == 0:
            assert got is out                       # equal shards land in the caller's tensor, in place
            q.put(out.numpy())
            with pytest.raises(ValueError):
                D.gather_scans(mine, dst=0, out=torch.zeros((11, 1, 180)))
        else:
            assert got is None
            q.put(None)
    finally:
        tdist.destroy_process_group()


def test_equal_shards_gather_in_place():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    out = [g for g in got if g is not None][0]
    want = np.concatenate([np.full((6, 1, 180), k + 1.0, dtype=np.float32) + np.arange(180, dtype=np.float32) / 1000.0
                           for k in range(2)])
    assert np.array_equal(out, want.astype(np.float32))


def _samples_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 5 if rank == 0 else 9                   # ranks log different numbers of rows
        X = torch.full((n, 1, 180), float(rank + 1)) + torch.arange(n, dtype=torch.float32)[:, None, None] / 100.0
        y = torch.arange(n, dtype=torch.float32)[:, None] + 10.0 * rank
        gx, gy = D.gather_samples(X, y, dst=0)
        q.put((gx.numpy(), gy.numpy()) if rank == 0 else (gx, gy))
    finally:
        tdist.destroy_process_group()


def test_samples_of_unequal_shards_reach_the_trainer_rank_in_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_samples_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sum(g == (None, None) for g in got if not isinstance(g[0], np.ndarray)) == 1
    gx, gy = [g for g in got if isinstance(g[0], np.ndarray)][0]
    assert gx.shape == (14, 1, 180) and gy.shape == (14, 1)
    assert np.array_equal(gy[:, 0], np.concatenate([np.arange(5.0), np.arange(9.0) + 10.0]).astype(np.float32))
    assert np.array_equal(gx[:, 0, 0], np.concatenate([1.0 + np.arange(5) / 100.0, 2.0 + np.arange(9) / 100.0]).astype(np.float32))
    with pytest.raises(ValueError):
        D.gather_samples(torch.zeros(3, 1, 180), torch.zeros(2, 1))
    x1, y1 = D.gather_samples(torch.zeros(3, 1, 180), torch.zeros(3, 1))       # no group: the tensors themselves
    assert x1.shape == (3, 1, 180) and y1.shape == (3, 1)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2`, not under torchrun, must come back as a two-rank run.  There is no GPU here, so
    the ranks rehearse everything but the simulation (AUV_BENCH_REHEARSE=nogpu): launch, rendezvous, barrier,
    the statistics all-reduce, the one JSON line of rank 0."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["AUV_BENCH_REHEARSE"] = "nogpu"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_joined"] == 2 and out["config"]["sharding"].startswith("envs x2")
    # a multi-rank line carries the two collectives of the 8-GPU configuration without being asked for them: the
    # statistics all-reduce and the gather of every rank's scans on the trainer rank, checked against their senders
    assert out["collectives"]["stats_allreduce_ms"] > 0
    assert out["collectives"]["scan_gather"]["checked"] is True and out["collectives"]["scan_gather"]["bytes_inbound"] > 0
    # a rank count that differs from --gpus is refused, not silently measured
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env2, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
