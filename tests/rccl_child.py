"""Child process of tests/test_rccl_gpu.py: a one-rank RCCL ("nccl") process
group on cuda:0, through which every collective of gym_auv_amd.dist is sent for
real, then destroyed.  Run in a process of its own so that the communicator's
device state never mixes with the test runner's.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    from gym_auv_amd import dist as D
    from gym_auv_amd import scenarios
    from tests.helpers import make_sim

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[1]}", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    dist.barrier()

    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=512)
    sim = make_sim(cfg, scene)
    sim.reset(scene["init_state"])
    acts = torch.from_numpy(scenarios.random_actions(512, 3, seed=1)).to(dev)
    for k in range(3):
        sim.step_device(acts[k].data_ptr())
    # statistics: accumulated on the device, all-reduced through RCCL
    st = D.EpisodeStats(dev)
    st.update_from_sim(sim)
    local = st.vec.clone()
    totals = st.reduce()
    out["on_device_equal"] = D.totals_as_dict(st.reduce_on_device()) == totals
    out["stats_equal"] = bool(torch.equal(torch.tensor([totals[k] for k in D.STAT_FIELDS], dtype=torch.float64), local.cpu()))
    out["env_steps"] = totals["env_steps"]
    t = torch.tensor([1.5, -2.0], dtype=torch.float64, device=dev)
    D.all_reduce(t, op=dist.ReduceOp.MAX)
    out["all_reduce_max"] = t.tolist()
    # scans: gathered into a caller's tensor and into a new one
    per = sim.tensor("perception")
    into = torch.zeros((512, 1, 180), dtype=torch.float32, device=dev)
    got = D.gather_scans(per, dst=0, out=into)
    out["gather_into_out"] = bool(got is into and torch.equal(into, per))
    out["gather_new"] = bool(torch.equal(D.gather_scans(per, dst=0), per))
    X, y = per[:100].clone(), torch.rand(100, 1, device=dev)
    gx, gy = D.gather_samples(X, y, dst=0)
    out["gather_samples"] = bool(torch.equal(gx, X) and torch.equal(gy, y))
    torch.cuda.synchronize(dev)
    sim.close()
    dist.barrier()
    dist.destroy_process_group()
    out["destroyed"] = not dist.is_initialized()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
