"""auv_episode_stats_static, the entry point that allocates nothing, against
auv_episode_stats: same bits, on two calls in a row, the second on a smaller
batch (the fixed scratch array then still holds the first call's records)."""
import ctypes as C

import numpy as np
import pytest

from gym_auv_amd import _lib, scenarios
from tests.helpers import make_sim

pytestmark = pytest.mark.gpu


def stepped_sim(name, V, seed):
    import torch

    cfg, scene = scenarios.make_workload(name, n_envs=V)
    cfg["max_steps"] = 2
    sim = make_sim(cfg, scene)
    st = torch.cuda.current_stream()
    sim.reset(scene["init_state"], stream=st)
    acts = torch.from_numpy(scenarios.random_actions(V, 3, seed=seed)).cuda()
    for k in range(3):
        sim.step_device(acts[k].data_ptr(), stream=st)
    st.synchronize()
    return sim


def both(sim, accumulate, start):
    """(existing entry point, allocation-free entry point) through the C ABI, from the same starting `out`."""
    import torch

    lib = sim._lib
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for fn in (lib.auv_episode_stats, lib.auv_episode_stats_static):
        out = torch.from_numpy(start.copy()).cuda()
        rc = fn(sim._h, C.c_void_p(out.data_ptr()), 1 if accumulate else 0, st)
        assert rc == 0, lib.auv_last_error(sim._h).decode()
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    return outs


def test_static_entry_point_equals_the_allocating_one_twice_in_a_row():
    big = stepped_sim("cfg3_65536_moving", 20000, seed=1)
    small = stepped_sim("cfg5_dense_720", 129, seed=2)
    start = np.arange(8, dtype=np.float64) * 0.5
    a, b = both(big, False, start)
    assert np.array_equal(a, b) and a[0] == 20000 and a[6] > 0
    # second call, smaller input, accumulating: records 3.. of the array are stale leftovers of the first call
    a2, b2 = both(small, True, start)
    assert np.array_equal(a2, b2) and a2[0] == 129 + start[0]
    # and back to the large batch
    a3, b3 = both(big, True, a)
    assert np.array_equal(a3, b3) and a3[0] == 40000
    big.close()
    small.close()


def test_static_entry_point_rejects_what_it_cannot_hold():
    import torch

    lib = _lib.load()
    assert _lib.STATS_MAX_ENVS == 1048576
    V = _lib.STATS_MAX_ENVS + 1
    cfg, scene = scenarios.make_workload("cfg2_4096_static", n_envs=8)
    cfg["n_envs"], cfg["n_circles"], cfg["n_rays"] = V, 1, 4
    from gym_auv_amd.sim import AuvSim

    sim = AuvSim(cfg)
    out = torch.zeros(8, dtype=torch.float64, device="cuda")
    rc = lib.auv_episode_stats_static(sim._h, C.c_void_p(out.data_ptr()), 0, None)
    assert rc == -1 and b"exceeds the fixed scratch array" in lib.auv_last_error(sim._h)
    assert lib.auv_episode_stats_static(sim._h, None, 0, None) == -1
    sim.close()
