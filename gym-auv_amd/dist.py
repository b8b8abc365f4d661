"""Multi-GPU layer: one process per GPU, envs sharded in contiguous blocks.

Stepping needs no communication (envs are independent).  Two collectives
exist, both outside the per-step path:
  * `reduce_episode_stats`: one all-reduce(sum) of a small statistics vector
    per rollout (64 B; latency-bound, so never per step);
  * `gather_scans`: the scans of every rank to one trainer rank.  A gather is
    peer-to-peer writes into the trainer, which on xGMI uses all of the
    trainer's inbound links in parallel; a ring all-gather would be bound by
    one link.
The backend is whatever the process group was created with: "nccl" (= RCCL
over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

STAT_FIELDS = ("env_steps", "episodes_done", "collisions", "goals", "truncations", "reward_sum",
               "min_range_sum", "risk_max_sum")


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous block [begin, end) of envs owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_total), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_scene(scene: dict, rank: int, world: int) -> dict:
    """This rank's slice of a full-size scene dict."""
    n = scene["path"].shape[0]
    b, e = shard_range(n, rank, world)
    return {k: (v[:, b:e] if k == "init_state" else v[b:e]) for k, v in scene.items()}


class EpisodeStats:
    """Running episode statistics of one rank, kept on the device of its
    tensors; `reduce()` returns the node-wide totals as a dict."""

    def __init__(self, device):
        self.vec = torch.zeros(len(STAT_FIELDS), dtype=torch.float64, device=device)

    def update(self, done, collision, reached_goal, truncated, reward, ranges=None, risk=None):
        """Accumulate one step's outputs (tensors of this rank's envs)."""
        n = done.numel()
        f64 = torch.float64
        parts = [torch.tensor(float(n), dtype=f64, device=self.vec.device), done.sum(dtype=f64),
                 collision.sum(dtype=f64), reached_goal.sum(dtype=f64), truncated.sum(dtype=f64),
                 reward.sum(dtype=f64),
                 ranges.amin(dim=-1).sum(dtype=f64) if ranges is not None else self.vec.new_zeros(()),
                 risk.amax(dim=-1).sum(dtype=f64) if risk is not None else self.vec.new_zeros(())]
        self.vec += torch.stack(parts)

    def update_from_sim(self, sim, stream=None, accumulate=True):
        """The same accumulation, reduced by the library itself (`auv_episode_stats`:
        two small launches, nothing allocated) instead of ten torch reductions.
        accumulate=False starts the sums over with this batch."""
        sim.episode_stats(self.vec, accumulate=accumulate, stream=stream)

    def reduce(self, group=None) -> dict:
        return reduce_episode_stats(self.vec, group)

    def reduce_on_device(self, group=None) -> torch.Tensor:
        """The node-wide totals as a device tensor, without waiting for them: the
        all-reduce is queued behind the kernels that formed the sums, and nothing
        is read back (`totals_as_dict` does that, whenever the caller synchronises)."""
        return all_reduce(self.vec.clone(), group=group)


def totals_as_dict(totals: torch.Tensor) -> dict:
    return dict(zip(STAT_FIELDS, totals.tolist()))


def all_reduce(t: torch.Tensor, op=dist.ReduceOp.SUM, group=None) -> torch.Tensor:
    """In-place all-reduce of a small tensor (a no-op without a process group;
    with one, the collective runs whatever the group's size, also on a single
    rank).  RCCL reduces device tensors in
    place; a gloo group (CPU tests, the one-GPU rehearsal of the multi-rank
    path) gets a host copy, because gloo's handling of device tensors stalls
    when several ranks share one GPU."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def reduce_episode_stats(vec: torch.Tensor, group=None) -> dict:
    """all-reduce(sum) of the statistics vector over all ranks."""
    return dict(zip(STAT_FIELDS, all_reduce(vec.clone(), group=group).tolist()))


def gather_scans(perception: torch.Tensor, dst: int = 0, group=None, out: torch.Tensor | None = None):
    """Gather every rank's scans `[V_local,1,R]` on rank `dst`, in rank order
    (= env order of the block sharding).  Returns `[V_total,1,R]` on `dst`,
    None elsewhere.  Equal shards (the benchmark's case) land directly in the
    slices of ONE result tensor (`out`, or a new one): no padding, no
    concatenation, nothing copied twice on the trainer.  Shards that differ in
    size by an env are padded to the longest for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return perception if out is None else out.copy_(perception)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = torch.zeros(world, dtype=torch.int64)
    sizes[rank] = perception.shape[0]
    if dist.get_backend(group) != "gloo":
        sizes = sizes.to(perception.device)
    dist.all_reduce(sizes, group=group)
    sizes = sizes.tolist()
    if perception.is_cuda and dist.get_backend(group) == "gloo":
        got = gather_scans(perception.cpu(), dst, group, None if out is None else out.cpu())
        if got is None:
            return None
        return got.to(perception.device) if out is None else out.copy_(got)
    vmax, total = max(sizes), sum(sizes)
    tail = tuple(perception.shape[1:])
    send = perception.contiguous()
    if min(sizes) == vmax:
        if rank != dst:
            dist.gather(send, None, dst=dst, group=group)
            return None
        if out is None:
            out = torch.empty((total,) + tail, dtype=perception.dtype, device=perception.device)
        elif tuple(out.shape) != (total,) + tail or out.dtype != perception.dtype or not out.is_contiguous():
            raise ValueError(f"gather_scans: out must be a contiguous {(total,) + tail} tensor of {perception.dtype}")
        dist.gather(send, list(out.split(vmax, dim=0)), dst=dst, group=group)   # views of `out`: written in place
        return out
    if perception.shape[0] < vmax:
        pad = perception.new_zeros((vmax - perception.shape[0],) + tail)
        send = torch.cat([perception, pad], dim=0).contiguous()
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=dst, group=group)
        return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)
    dist.gather(send, None, dst=dst, group=group)
    return None


def gather_samples(X: torch.Tensor, y: torch.Tensor, dst: int = 0, group=None):
    """The training samples every rank collected from its envs
    (`dataset.DeviceDatasetLogger.samples()`: X `[n_local,1,R]`, y `[n_local,1]`)
    on the trainer rank `dst`, in rank order; ranks hold different numbers of
    rows.  Returns (X, y) on `dst`, (None, None) elsewhere; what the trainer
    rank then hands to `trainer.DeviceTrainer`."""
    if len(X) != len(y):
        raise ValueError("X and y differ in length")
    return gather_scans(X, dst, group), gather_scans(y, dst, group)
