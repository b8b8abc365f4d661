"""Inputs the fast paths cannot serve must be routed or refused, never served wrongly."""
import numpy as np
import pytest

from gym_auv_amd import _lib, cnn, params, scenarios
from gym_auv_amd.sim import AuvSim
from oracle import auv_oracle as O
from oracle import cnn_oracle as N
from tests.helpers import PAIRS, make_sim, same_bits

pytestmark = pytest.mark.gpu


def sector_table(n_rays, lo=-1.0, hi=1.0):
    ang = np.linspace(lo, hi, n_rays)
    return np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang)], axis=1))


@pytest.mark.parametrize("table", ["sector", "turned", "uneven"])
def test_custom_ray_table_takes_the_all_pairs_path_and_matches_the_oracle(table):
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=300)
    R = cfg["n_rays"]
    if table == "sector":
        rays = sector_table(R)
    elif table == "turned":
        ang = -np.pi + (np.arange(R) + 1.5) * (2 * np.pi / R)      # half a ray off the standard layout
        rays = np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang)], axis=1))
    else:
        ang = np.sort(np.random.default_rng(5).uniform(-np.pi, np.pi, R))
        rays = np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang)], axis=1))
    sim = AuvSim(cfg, ray_dirs=rays)
    with pytest.raises(_lib.AuvError, match="evenly spaced"):
        sim.set_option(_lib.OPT_LIDAR_MODE, _lib.LIDAR_RASTER)
    with pytest.raises(_lib.AuvError):
        sim.set_option(_lib.OPT_AUTO_RESET, 1)
    sim.set_scene(scene)
    env = O.OracleEnv(cfg, rays)
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"], scene["poly_heading"])
    env.set_path(scene["path"])
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    acts = scenarios.random_actions(300, 3, seed=2)
    for k in range(3):
        sim.step_host(acts[k])
        env.step(acts[k])
        for name, attr in PAIRS.items():
            got = sim.read(name)
            assert same_bits(got, np.asarray(getattr(env, attr)).reshape(got.shape)), (name, k)
    assert (env.hit >= 0).any()
    with pytest.raises(_lib.AuvError, match="ray table"):      # the refusal names its cause
        sim.set_option(_lib.OPT_AUTO_RESET, 1)
    # the standard table again: the batch is back on the rasterising path by itself (the custom table had forced
    # the other one, the caller had not chosen it), so the in-step restart is accepted again
    sim.set_rays(params.ray_table(R))
    sim.set_option(_lib.OPT_AUTO_RESET, 1)
    sim.set_option(_lib.OPT_AUTO_RESET, 0)
    # an all-pairs path chosen by the caller stays chosen whatever table is uploaded
    sim.set_option(_lib.OPT_LIDAR_MODE, _lib.LIDAR_ALL_PAIRS)
    sim.set_rays(rays)
    sim.set_rays(params.ray_table(R))
    with pytest.raises(_lib.AuvError, match="AUV_OPT_LIDAR_MODE selected it"):
        sim.set_option(_lib.OPT_AUTO_RESET, 1)
    sim.set_option(_lib.OPT_LIDAR_MODE, _lib.LIDAR_RASTER)
    sim.close()


def test_custom_rays_refused_while_in_step_restart_is_on():
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=16)
    sim = AuvSim(cfg)
    sim.set_option(_lib.OPT_AUTO_RESET, 1)
    with pytest.raises(_lib.AuvError, match="in-step restart"):
        sim.set_rays(sector_table(cfg["n_rays"]))
    sim.close()


def test_partial_scene_upload_is_not_a_complete_scene():
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=40)
    sim = AuvSim(cfg)
    tail = {k: (v[:, 30:] if k == "init_state" else v[30:]) for k, v in scene.items()}
    sim.set_scene(tail, env_begin=30)                 # only the last shard
    with pytest.raises(_lib.AuvError, match=r"envs \[0, 40\) are missing"):
        sim.reset(scene["init_state"])
    head = {k: (v[:, :30] if k == "init_state" else v[:30]) for k, v in scene.items()}
    sim.set_scene(head, env_begin=0)
    with pytest.raises(_lib.AuvError, match=r"envs \[30, 40\) are missing"):
        sim.reset(scene["init_state"])
    sim.set_scene(tail, env_begin=30)
    sim.reset(scene["init_state"])                    # front to back: complete
    sim.close()


def test_oversized_tile_request_is_cut_down_to_what_fits():
    cfg, scene = scenarios.make_workload("cfg5_dense_720", n_envs=64)
    sim = AuvSim(cfg)
    sim.set_option(_lib.OPT_RASTER_TILE, 64)          # 64 envs x 1,152 primitives cannot share one LDS image
    sim.set_scene(scene)
    sim.reset(scene["init_state"])                    # launches with the largest tile that fits
    env = O.OracleEnv(cfg, params.ray_table(cfg["n_rays"]))
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"], scene["poly_heading"])
    env.set_path(scene["path"])
    env.reset(scene["init_state"])
    assert same_bits(sim.read("ranges"), env.ranges) and same_bits(sim.read("hit"), env.hit)
    with pytest.raises(_lib.AuvError):
        sim.set_option(_lib.OPT_RASTER_TILE, 65)
    sim.close()


def test_cnn_forward_checks_its_output_tensor():
    import os

    import torch

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnn_golden.npz"))
    net = cnn.LidarCNN("shallow", N.golden_weights(gold, "shallow"))
    x = torch.rand(5, 1, 180, device="cuda")
    for bad in (torch.empty(5, 1, dtype=torch.float64, device="cuda"), torch.empty(4, 1, device="cuda"),
                torch.empty(5, 12, device="cuda"), torch.empty(5, 2, device="cuda")[:, :1], torch.empty(5, 1)):
        with pytest.raises(ValueError):
            net.forward(x, "risk", out=bad)
    assert net.forward(x, "risk", out=torch.empty(5, 1, device="cuda")).shape == (5, 1)
    net.close()


def test_a_rollout_can_be_captured_into_a_graph_and_replays_bit_for_bit():
    """Stepping with resident actions allocates nothing and never synchronises, so a rollout can be captured on a
    stream (HIP graph) as it is; replaying the graph must leave every buffer as plain stepping leaves it, episodes
    restarting inside the steps included."""
    import torch

    from gym_auv_amd.sim import AuvSim

    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=1000)
    cfg["max_steps"] = 17                       # every episode ends inside the rollout and restarts on the device
    acts = torch.from_numpy(scenarios.random_actions(1000, 8, seed=5)).cuda()
    ptrs = [acts[i].data_ptr() for i in range(8)]
    K, got = 40, {}
    for mode in ("plain", "graph"):
        sim = AuvSim(cfg)
        sim.set_scene(scene)
        sim.set_option(_lib.OPT_AUTO_RESET, 1)
        st = torch.cuda.current_stream()
        if mode == "graph":
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):       # a first launch outside the capture
                sim.reset(scene["init_state"], stream=side)
                sim.step_device(ptrs[0], stream=side)
            torch.cuda.synchronize()
        sim.reset(scene["init_state"], stream=st)
        torch.cuda.synchronize()
        if mode == "plain":
            for k in range(K):
                sim.step_device(ptrs[k % 8], stream=st)
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                s = torch.cuda.current_stream()
                for k in range(K):
                    sim.step_device(ptrs[k % 8], stream=s)
            assert int(sim.read("t_step").max()) == 0      # capturing executed nothing
            g.replay()
        torch.cuda.synchronize()
        got[mode] = {k: sim.read(k).copy() for k in ("state", "ranges", "hit", "perception", "obst_dist", "meta", "risk",
                                                     "reward", "done", "t_step", "circles", "poly_pose")}
        sim.close()
    assert int(got["plain"]["t_step"].max()) < K           # restarts happened
    for k, v in got["plain"].items():
        assert np.array_equal(v, got["graph"][k], equal_nan=True), k


def test_paths_with_a_repeated_waypoint_or_a_bad_direction_are_refused():
    """A vessel leaves a segment when its projection exceeds the segment's length; a segment of length 0 (repeated
    waypoint) or without a unit direction would hold it for ever, so the upload refuses both.  Slots beyond the
    segments in use are not looked at."""
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=8)
    cfg["n_path_segments"] = 3
    sim = AuvSim(cfg)
    sim.set_scene(scene)
    pts = np.zeros((8, 4, 2))
    pts[:, 1] = (100.0, 0.0)
    pts[:, 2] = (200.0, 50.0)
    pts[:, 3] = (300.0, 50.0)
    seg, n = scenarios.waypoint_path(pts)
    sim.set_path(seg, n)
    twice = pts.copy()
    twice[5, 2] = twice[5, 1]                      # env 5 names a waypoint twice
    seg2, n2 = scenarios.waypoint_path(twice)
    with pytest.raises(_lib.AuvError, match=r"env 5, segment 1: length 0"):
        sim.set_path(seg2, n2)
    n_short = n2.copy()
    n_short[5] = 1                                 # ... but uses only the first segment: fine
    sim.set_path(seg2, n_short)
    bad = seg.copy()
    bad[2, 0, 3:5] = (0.6, 0.6)                    # not (cos, sin) of an angle
    with pytest.raises(_lib.AuvError, match=r"env 2, segment 0"):
        sim.set_path(bad, n)
    sim.close()


def test_launch_info_names_the_shape_the_sensing_kernel_runs_in():
    """`auv_launch_info`: the benchmark's shape takes tiles of 4 envs on 256 threads with a 19.4 KiB LDS image, and a
    compute unit holds 8 such workgroups (every wave slot it has); forced shapes and an attached network show."""
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=512)
    sim = make_sim(cfg, scene)
    small = sim.launch_info()          # a batch this small is spread over the device: one env per workgroup
    assert small["envs_per_workgroup"] == 1 and small["threads"] == 128
    sim.set_option(_lib.OPT_RASTER_TILE, 4)      # what a batch of 65,536 gets by itself
    sim.set_option(_lib.OPT_SENSE_BLOCK, 256)
    info = sim.launch_info()
    assert (info["envs_per_workgroup"], info["threads"]) == (4, 256)
    assert info["lds_bytes"] <= 160 * 1024 // 8 and info["workgroups_per_cu"] == 8
    sim.set_option(_lib.OPT_RASTER_TILE, 8)
    sim.set_option(_lib.OPT_SENSE_BLOCK, 512)
    wide = sim.launch_info()
    assert (wide["envs_per_workgroup"], wide["threads"]) == (8, 512) and wide["lds_bytes"] > info["lds_bytes"]
    assert 1 <= wide["workgroups_per_cu"] <= 4
    sim.close()
    cfg, scene = scenarios.make_workload("cfg5_dense_720", n_envs=8)
    sim = make_sim(cfg, scene)
    dense = sim.launch_info()
    assert dense["envs_per_workgroup"] == 1 and dense["threads"] == 768 and dense["workgroups_per_cu"] == 2
    sim.close()

