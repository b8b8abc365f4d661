"""auv_episode_stats: the on-device reduction of a batch's outputs."""
import numpy as np
import pytest

from gym_auv_amd import dist as D
from gym_auv_amd import scenarios
from tests.helpers import make_sim

pytestmark = pytest.mark.gpu


def host_stats(sim):
    done, coll = sim.read("done"), sim.read("collision")
    goal, trunc = sim.read("reached_goal"), sim.read("truncated")
    risk = sim.read("risk")
    return np.array([len(done), done.sum(), coll.sum(), goal.sum(), trunc.sum(), sim.read("reward").sum(),
                     sim.read("ranges").min(axis=1).sum(), risk.max(axis=1).sum() if risk.shape[1] else 0.0])


@pytest.mark.parametrize("name,V", [("cfg3_65536_moving", 5000), ("cfg3_65536_moving", 63), ("cfg2_4096_static", 4096),
                                    ("cfg5_dense_720", 130), ("cfg1_single_episode", 1)])
def test_episode_stats_equal_host_sums(name, V):
    import torch

    cfg, scene = scenarios.make_workload(name, n_envs=V)
    cfg["max_steps"] = 3   # some envs truncate within the rollout
    sim = make_sim(cfg, scene)
    st = torch.cuda.current_stream()
    sim.reset(scene["init_state"], stream=st)
    acts = torch.from_numpy(scenarios.random_actions(V, 5, seed=3)).cuda()
    out = torch.full((8,), -7.0, dtype=torch.float64, device="cuda")
    total = np.zeros(8)
    ref = D.EpisodeStats(torch.device("cuda"))
    for k in range(5):
        sim.step_device(acts[k].data_ptr(), stream=st)
        sim.episode_stats(out, accumulate=(k > 0), stream=st)
        ref.update(sim.tensor("done"), sim.tensor("collision"), sim.tensor("reached_goal"), sim.tensor("truncated"),
                   sim.tensor("reward"), ranges=sim.tensor("ranges"), risk=sim.tensor("risk"))
        st.synchronize()
        total += host_stats(sim)
    got = out.cpu().numpy()
    assert np.array_equal(got[:5], total[:5])                      # counts are exact
    assert np.allclose(got[5:], total[5:], rtol=1e-12, atol=1e-9)  # fp64 sums in another order
    assert np.allclose(got, ref.vec.cpu().numpy(), rtol=1e-12, atol=1e-9)
    assert got[4] > 0
    # same bits on every call
    again = torch.empty(8, dtype=torch.float64, device="cuda")
    once = torch.empty(8, dtype=torch.float64, device="cuda")
    sim.episode_stats(once, stream=st)
    sim.episode_stats(again, stream=st)
    assert torch.equal(once, again)
    # the accumulating wrapper
    es = D.EpisodeStats(torch.device("cuda"))
    es.update_from_sim(sim, stream=st)
    assert torch.equal(es.vec, once)
    sim.close()


def test_episode_stats_argument_checks():
    import torch

    cfg, scene = scenarios.make_workload("cfg2_4096_static", n_envs=8)
    sim = make_sim(cfg, scene)
    with pytest.raises(Exception):
        sim.episode_stats(torch.zeros(8, dtype=torch.float64, device="cuda"))   # before the first reset
    sim.reset(scene["init_state"])
    with pytest.raises(TypeError):
        sim.episode_stats(torch.zeros(8, dtype=torch.float32, device="cuda"))
    with pytest.raises(TypeError):
        sim.episode_stats(torch.zeros(7, dtype=torch.float64, device="cuda"))
    sim.close()


def test_a_nan_in_the_ranges_or_the_risks_shows_in_the_sums():
    """torch's amin / amax hand a NaN on, and so does the device reduction: a NaN planted in one env's risks and in
    another env's ranges makes both sums NaN on both paths, and nothing else."""
    import torch

    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=200)
    sim = make_sim(cfg, scene)
    sim.reset(scene["init_state"])
    dev = torch.device("cuda")
    clean = torch.zeros(8, dtype=torch.float64, device=dev)
    sim.episode_stats(clean)
    assert bool(torch.isfinite(clean).all())
    sim.tensor("risk")[77, 3] = float("nan")
    sim.tensor("ranges")[131, 179] = float("nan")
    got = torch.zeros(8, dtype=torch.float64, device=dev)
    sim.episode_stats(got)
    ref = D.EpisodeStats(dev)
    ref.update(sim.tensor("done"), sim.tensor("collision"), sim.tensor("reached_goal"), sim.tensor("truncated"),
               sim.tensor("reward"), ranges=sim.tensor("ranges"), risk=sim.tensor("risk"))
    assert torch.equal(torch.isnan(got), torch.isnan(ref.vec))
    assert torch.isnan(got).tolist() == [False] * 6 + [True, True]
    assert torch.equal(got[:6], clean[:6])
    sim.close()
