"""HIP path vs the CPU oracle, through the C ABI, on the same seeded inputs.

The bar is bit-exact for EVERY output (hit ids, flags, and also the fp64
state, ranges, distances, metadata, risk and the fp32 observations): both
sides follow one unfused-fp64 arithmetic contract, so any difference is a
bug, not rounding.  The looser tolerances north_star allows (state 1e-9,
ranges 1e-6 relative) are therefore never needed.  There is no reference
fixture for these outputs (the simulator is not in the reference checkout):
PARITY UNPINNED beyond this oracle.
"""
import numpy as np
import pytest

from gym_auv_amd import params, scenarios
from gym_auv_amd.sim import AuvSim
from oracle import auv_oracle as O
from tests.helpers import PAIRS, assert_all_equal, make_oracle, make_sim, same_bits

pytestmark = pytest.mark.gpu

# both LiDAR kernels must give the oracle's bits: 0 = rasterising (default), 1 = all pairs
MODES = pytest.mark.parametrize("mode", [0, 1], ids=["raster", "allpairs"])


def _run_pair(cfg, scene, steps, seed, check_every=1, mode=None):
    sim = make_sim(cfg, scene, lidar_mode=mode)
    env = make_oracle(cfg, scene)
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    assert_all_equal(sim, env, "after reset")
    acts = scenarios.random_actions(cfg["n_envs"], steps, seed=seed)
    for k in range(steps):
        sim.step_host(acts[k])
        env.step(acts[k])
        if (k + 1) % check_every == 0 or k == steps - 1:
            assert_all_equal(sim, env, f"after step {k + 1}")
    sim.close()


@MODES
def test_cfg1_single_vessel_episode(mode):
    cfg, scene = scenarios.make_workload("cfg1_single_episode")
    _run_pair(cfg, scene, steps=300, seed=0, check_every=10, mode=mode)


@MODES
def test_cfg2_static_circles(mode):
    cfg, scene = scenarios.make_workload("cfg2_4096_static", n_envs=300)
    _run_pair(cfg, scene, steps=25, seed=1, check_every=5, mode=mode)


@MODES
def test_cfg3_moving_polygons(mode):
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=333)
    _run_pair(cfg, scene, steps=25, seed=2, check_every=5, mode=mode)


@MODES
def test_cfg5_dense_720_rays(mode):
    cfg, scene = scenarios.make_workload("cfg5_dense_720", n_envs=6)
    _run_pair(cfg, scene, steps=3, seed=4, mode=mode)


@MODES
def test_empty_scene_and_absent_slots(mode):
    # no obstacle slots at all
    cfg = params.make_config(70, 0, 0)
    scene = scenarios.make_scene(70, 0, 0, seed=11)
    _run_pair(cfg, scene, steps=3, seed=11, mode=mode)
    # slots present but switched off (radius <= 0, nv == 0), ragged vertex counts
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=65)
    scene["circles"][::2, 1::2, 2] = 0.0
    scene["circles"][1::3, 0, 2] = -1.0
    scene["poly_nv"][::3, ::2] = 0
    scene["poly_nv"][1::3, 1] = 3
    scene["poly_nv"][2::3, 2] = 4
    _run_pair(cfg, scene, steps=5, seed=12, mode=mode)


def test_collisions_goal_and_truncation_flags():
    # tight scene so that vessels do hit things, short path so that some arrive
    cfg = params.make_config(128, 6, 6, 5, max_steps=30, goal_radius=5.0)
    scene = scenarios.make_scene(128, 6, 6, 5, seed=21, extent=60.0, width=(8.0, 25.0))
    scene["path"][:, 5] = 12.0
    sim = make_sim(cfg, scene)
    env = make_oracle(cfg, scene)
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    acts = scenarios.random_actions(128, 40, seed=22)
    acts[:, 0] = 1.0
    acts[:, 1] *= 0.1
    seen = np.zeros(3, dtype=bool)
    for k in range(40):
        sim.step_host(acts[k])
        env.step(acts[k])
        assert_all_equal(sim, env, f"after step {k + 1}")
        seen |= [env.collision.any(), env.reached_goal.any(), env.truncated.any()]
    assert seen.all(), f"scenario did not exercise collision/goal/truncation: {seen}"
    sim.close()


def test_in_step_restart_matches_oracle():
    """AUV_OPT_AUTO_RESET: an env whose done flag is set restarts inside the next
    step (initial state, step counter 0, obstacles at their uploaded poses,
    reward 0), with no host round trip.  Episodes end here by collision, goal
    and truncation, repeatedly, and every buffer must follow the oracle's bits;
    the batch does not fill its last tile."""
    from gym_auv_amd import _lib

    V = 131
    cfg = params.make_config(V, 6, 6, 5, max_steps=30, goal_radius=5.0)
    scene = scenarios.make_scene(V, 6, 6, 5, seed=25, extent=60.0, width=(8.0, 25.0))
    scene["path"][:, 5] = np.array([12.0, 25.0, 60.0])[np.arange(V) % 3]     # short paths end at the goal, long ones run out of steps
    sim = make_sim(cfg, scene)
    sim.set_option(_lib.OPT_AUTO_RESET, 1)
    env = make_oracle(cfg, scene)
    env.auto_reset = True
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    acts = scenarios.random_actions(V, 100, seed=24)
    acts[:, 0] = 1.0
    acts[:, 1] *= 0.1
    restarts = 0
    seen = np.zeros(3, dtype=bool)
    for k in range(100):
        restarts += int(env.done.sum())
        sim.step_host(acts[k])
        env.step(acts[k])
        assert_all_equal(sim, env, f"after step {k + 1}")
        seen |= [env.collision.any(), env.reached_goal.any(), env.truncated.any()]
    assert restarts > V and seen.all(), (restarts, seen)
    # the cross-check path has no in-step restart and says so
    with pytest.raises(Exception, match="all-pairs"):
        sim.set_option(_lib.OPT_LIDAR_MODE, _lib.LIDAR_ALL_PAIRS)
    sim.close()


def test_generated_scenes_and_regeneration_match_oracle():
    """The device-side scenario generator against its numpy twin: the generated
    scene itself, then 100 steps in which finished episodes restart in-step with
    a NEW scene for their next episode number (no host round trip); every
    buffer, the live obstacle records and the episode counters must agree."""
    from gym_auv_amd import _lib

    V = 133
    cfg = params.make_config(V, 5, 6, 6, max_steps=20, goal_radius=5.0)
    gp = dict(extent=70.0, w_lo=8.0, w_hi=25.0, path_length=14.0, clearance=4.0)
    sim = AuvSim(cfg)
    env = O.OracleEnv(cfg, params.ray_table(cfg["n_rays"]))
    sim.generate_scene(0x9E3779B97F4A7C15, gp)
    env.generate_scene(0x9E3779B97F4A7C15, gp)
    sim.reset(None)
    env.reset(None)
    assert_all_equal(sim, env, "after generation + reset")
    sim.set_option(_lib.OPT_AUTO_RESET, 1)
    sim.set_option(_lib.OPT_REGENERATE, 1)
    env.auto_reset = env.regenerate = True
    acts = scenarios.random_actions(V, 100, seed=26)
    acts[:, 0] = 1.0          # full thrust, free rudder: some vessels reach the goal in 20 steps, some do not
    seen = np.zeros(3, dtype=bool)
    for k in range(100):
        sim.step_host(acts[k])
        env.step(acts[k])
        assert_all_equal(sim, env, f"after step {k + 1}")
        seen |= [env.collision.any(), env.reached_goal.any(), env.truncated.any()]
    assert env.episode.min() >= 1 and env.episode.max() >= 4 and seen.all(), (env.episode.min(), env.episode.max(), seen)
    # defaults (params == NULL) and a second seed give the oracle's scene too
    sim.generate_scene(5)
    env.generate_scene(5)
    sim.reset(None)
    env.reset(None)
    assert_all_equal(sim, env, "default parameters")
    # the size law of the reference's logged scenes: whole-number radii and ship sizes, bell-shaped
    sim.generate_scene(11, scenarios.REFERENCE_LIKE)
    env.generate_scene(11, scenarios.REFERENCE_LIKE)
    sim.reset(None)
    env.reset(None)
    assert_all_equal(sim, env, "reference-like parameters")
    rad = sim.read("circles")[:, :, 2]
    assert np.array_equal(rad, np.rint(rad)) and rad.min() >= 15 and rad.max() <= 41
    for k in range(3):
        sim.step_host(acts[k])
        env.step(acts[k])
    assert_all_equal(sim, env, "reference-like parameters, stepped and restarted")
    sim.close()
    # ships need five vertex slots
    sim = AuvSim(params.make_config(4, 1, 1, 4))
    with pytest.raises(Exception, match="n_verts"):
        sim.generate_scene(1)
    sim.close()


@pytest.mark.parametrize("risk_model", [1, 0], ids=["reference", "cpa"])
def test_metadata_refresh_interval(risk_model):
    """AUV_OPT_META_EVERY = 5: bearing, course and risk (and, with the reference's
    index, the heading and speed columns that follow its conventions) follow the
    oracle on every fifth step and keep their values in between; everything else
    (state, scan, distances, flags, reward, the other metadata fields) is current
    on every step."""
    from gym_auv_amd import _lib

    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=203)
    cfg = dict(cfg, risk_model=risk_model)
    held_fields = [0, 2, 3, 4] if risk_model == 1 else [3, 4]
    sim = make_sim(cfg, scene)
    env = make_oracle(cfg, scene)
    sim.set_option(_lib.OPT_META_EVERY, 5)
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    assert_all_equal(sim, env, "after reset")
    acts = scenarios.random_actions(203, 12, seed=9)
    held_meta, held_risk = env.meta.copy(), env.risk.copy()
    for k in range(12):
        sim.step_host(acts[k])
        env.step(acts[k])
        if (k + 1) % 5 == 0:
            held_meta, held_risk = env.meta.copy(), env.risk.copy()
        want_meta = env.meta.copy()
        want_meta[..., held_fields] = held_meta[..., held_fields]
        assert same_bits(sim.read("meta"), want_meta), k
        assert same_bits(sim.read("risk"), held_risk), k
        for name in ("state", "ranges", "hit", "perception", "navigation", "obst_dist", "collision", "done",
                     "reward", "min_dist", "min_idx", "circles", "poly_pose"):
            assert same_bits(sim.read(name), np.asarray(getattr(env, name)).reshape(sim.read(name).shape)), (name, k)
    sim.sense()   # a sensing pass outside a step refreshes everything
    assert same_bits(sim.read("meta"), env.meta) and same_bits(sim.read("risk"), env.risk)
    sim.close()


def test_full_size_restarts_with_regeneration_against_oracle_sample():
    """65,536 vessels with short episodes (truncation after 20 steps, tight
    generated scenes): every env restarts in-step several times, each time with
    a new generated scene.  The numpy oracle follows a random sample of 160 env
    numbers (the generator is a function of the env number, envs are
    independent): every buffer of the sample must match after every step."""
    from gym_auv_amd import _lib

    V, n_pick, steps = 65536, 160, 70
    cfg = params.make_config(V, 8, 8, 5, max_steps=20)
    gp = dict(extent=100.0, clearance=3.0, w_lo=8.0, w_hi=30.0, path_length=30.0)
    pick = np.sort(np.random.default_rng(9).choice(V, n_pick, replace=False))
    sim = AuvSim(cfg)
    sim.generate_scene(77, gp)
    sim.set_option(_lib.OPT_AUTO_RESET, 1)
    sim.set_option(_lib.OPT_REGENERATE, 1)
    sim.reset(None)
    env = O.OracleEnv(dict(cfg, n_envs=n_pick), params.ray_table(cfg["n_rays"]))
    env.generate_scene(77, gp, env_ids=pick)
    env.auto_reset = env.regenerate = True
    env.reset(None)
    acts = scenarios.random_actions(V, steps, seed=10)
    acts[:, 0] = 1.0
    ended_early = 0   # episodes of the whole batch seen ending by contact or arrival, not by truncation
    for k in range(steps):
        sim.step_host(acts[k])
        env.step(acts[k][:, pick])
        if k % 3 == 2 or k == steps - 1:
            ended_early += int((sim.read("collision") | sim.read("reached_goal")).sum())
            for name, attr in PAIRS.items():
                got = sim.read(name)
                got = got[:, pick] if name == "state" else got[pick]
                want = np.asarray(getattr(env, attr)).reshape(got.shape)
                assert same_bits(got, want), (name, k)
    ep = sim.read("episode")
    assert ep.min() >= 3 and ended_early > 0
    sim.close()


def test_masked_reset_matches_oracle():
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=97)
    sim = make_sim(cfg, scene)
    env = make_oracle(cfg, scene)
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    acts = scenarios.random_actions(97, 8, seed=31)
    for k in range(4):
        sim.step_host(acts[k])
        env.step(acts[k])
    mask = np.random.default_rng(5).random(97) < 0.3
    sim.reset(scene["init_state"], mask=mask)
    env.reset(scene["init_state"], mask=mask)
    assert_all_equal(sim, env, "after masked reset")
    for k in range(4, 8):
        sim.step_host(acts[k])
        env.step(acts[k])
    assert_all_equal(sim, env, "after steps following a masked reset")
    sim.close()


def _adversarial_scene(V, seed):
    """Geometry that stresses the candidate intervals of the rasterising
    kernel: vessels inside / touching / grazing obstacles, tiny and huge
    circles, slivers, edges through the vessel, obstacles straddling the
    sensor range and the scan's wrap-around (dead astern)."""
    rng = np.random.default_rng(seed)
    C, P, NV = 6, 6, 6
    cfg = params.make_config(V, C, P, NV)
    scene = scenarios.make_scene(V, C, P, NV, seed=seed, extent=170.0, n_sides=6, width=(0.5, 60.0))
    circ = scene["circles"]
    circ[:, 0, 2] = rng.choice([1e-6, 1e-3, 0.05, 400.0], V)           # pin-points and giants
    circ[:, 1, :2] = rng.normal(0.0, 3.0, (V, 2))                      # on top of the vessel
    circ[:, 1, 2] = rng.uniform(0.5, 8.0, V)
    ang = rng.uniform(-np.pi, np.pi, V)
    rad = rng.uniform(5.0, 30.0, V)
    d = 150.0 + rad * rng.choice([-1.0, -0.999999, 0.0, 0.999999, 1.0, 1.000001], V)
    circ[:, 2, 0], circ[:, 2, 1], circ[:, 2, 2] = d * np.cos(ang), d * np.sin(ang), rad   # range edge
    scene["poly_pose"][:, 0, :2] = rng.normal(0.0, 2.0, (V, 2))        # polygon around / through the vessel
    scene["poly_pose"][:, 1, :2] = rng.normal(0.0, 40.0, (V, 2))
    scene["poly_verts"][:, 2] *= np.array([1.0, 1e-4])                 # sliver
    scene["poly_pose"][::7, 3, 2:4] = [1.0, 0.0]                       # unrotated, so that ...
    scene["poly_verts"][::7, 3, 0] = -scene["poly_pose"][::7, 3, :2]   # ... vertex 0 sits exactly on the vessel
    scene["init_state"][2, ::5] = rng.choice([-np.pi, np.pi - 1e-12, 0.0, np.pi / 2], len(range(0, V, 5)))
    return cfg, scene


def test_raster_equals_all_pairs_on_adversarial_geometry():
    from gym_auv_amd import _lib

    V = 20000
    for seed in (101, 102):
        cfg, scene = _adversarial_scene(V, seed)
        out = []
        for mode in (_lib.LIDAR_RASTER, _lib.LIDAR_ALL_PAIRS):
            sim = make_sim(cfg, scene, lidar_mode=mode)
            sim.reset(scene["init_state"])
            acts = scenarios.random_actions(V, 3, seed=seed)
            res = [(sim.read("ranges"), sim.read("hit"))]
            for k in range(3):
                sim.step_host(acts[k])
                res.append((sim.read("ranges"), sim.read("hit")))
            out.append(res)
            sim.close()
        for (r0, h0), (r1, h1) in zip(*out):
            assert same_bits(r0, r1) and same_bits(h0, h1)
        assert (out[0][0][1] >= 0).mean() > 0.2          # the scenes do produce hits
    # and the oracle agrees on a slice of the same scenes
    sub = {k: (v[:, :150] if k == "init_state" else v[:150]) for k, v in scene.items()}
    sub_cfg = dict(cfg, n_envs=150)
    _run_pair(sub_cfg, sub, steps=2, seed=7, mode=_lib.LIDAR_RASTER)


def test_soak_raster_equals_all_pairs_on_generated_scenes():
    """Volume check of the rasteriser's culling: two batches of 65,536 vessels,
    one per LiDAR kernel, run in lockstep through 12 generated scene sets (crowded
    to sparse; vessels graze, enter and leave hulls) x 60 steps; ranges, hit ids,
    distances, metadata and flags are compared on the device after every step
    (8.5e9 rays in all)."""
    import torch

    from gym_auv_amd import _lib

    cfg, _ = scenarios.make_workload("cfg3_65536_moving")
    V = cfg["n_envs"]
    sims = [AuvSim(cfg), AuvSim(cfg)]
    sims[1].set_option(_lib.OPT_LIDAR_MODE, _lib.LIDAR_ALL_PAIRS)
    st = torch.cuda.current_stream()
    names = ("ranges", "hit", "obst_dist", "collision", "done", "min_idx", "risk", "meta", "reward")
    views = [[s.tensor(n) for n in names] for s in sims]
    acts = torch.from_numpy(scenarios.random_actions(V, 60, seed=5)).cuda()
    acts[:, 0] = 1.0
    n_hits = 0
    for seed in range(12):
        # from crowded (hulls from half a metre) to sparse, small to giant obstacles, slow to fast ships
        gp = dict(extent=(60.0, 120.0, 200.0, 400.0)[seed % 4], clearance=(0.5, 5.0, 25.0)[seed % 3],
                  r_hi=(30.0, 80.0)[seed % 2], w_lo=8.0, w_hi=(20.0, 40.0)[(seed // 2) % 2], s_hi=(1.2, 4.0)[seed % 2])
        for s in sims:
            s.generate_scene(1000 + seed, gp, stream=st)
            s.reset(None, stream=st)
        for k in range(60):
            for s in sims:
                s.step_device(acts[k].data_ptr(), stream=st)
            for n, x, y in zip(names, views[0], views[1]):
                assert torch.equal(x, y), (seed, k, n)
        n_hits += int((views[0][1] >= 0).sum())
    assert n_hits > 1000000    # the scenes are not empty
    for s in sims:
        s.close()


def test_full_size_batch_against_oracle_sample():
    """cfg3 at its full 65,536 vessels: structural invariants on every ray and
    the oracle on a random sample of envs (each env is independent, so a
    sub-batch of the same envs must reproduce the full batch bit for bit)."""
    cfg, scene = scenarios.make_workload("cfg3_65536_moving")
    V = cfg["n_envs"]
    sim = make_sim(cfg, scene)
    sim.reset(scene["init_state"])
    acts = scenarios.random_actions(V, 3, seed=41)
    for k in range(3):
        sim.step_host(acts[k])
    ranges, hit = sim.read("ranges"), sim.read("hit")
    assert ((ranges > 0) & (ranges <= 150.0)).all()
    assert np.array_equal(hit == -1, ranges == 150.0)
    assert hit.max() < cfg["n_circles"] + cfg["n_polys"]
    assert same_bits(sim.read("perception")[:, 0], (1.0 - ranges / 150.0).astype(np.float32))
    dist = sim.read("obst_dist")
    assert np.array_equal(sim.read("collision"), (dist.min(axis=1) < 0).astype(np.uint8))
    pick = np.sort(np.random.default_rng(6).choice(V, 200, replace=False))
    sub_cfg = dict(cfg, n_envs=len(pick))
    sub_scene = {k: (v[:, pick] if k == "init_state" else v[pick]) for k, v in scene.items()}
    env = make_oracle(sub_cfg, sub_scene)
    env.reset(sub_scene["init_state"])
    for k in range(3):
        env.step(acts[k][:, pick])
    assert same_bits(ranges[pick], env.ranges)
    assert same_bits(hit[pick], env.hit)
    assert same_bits(sim.read("state")[:, pick], env.state)
    assert same_bits(dist[pick], env.obst_dist)
    assert same_bits(sim.read("risk")[pick], env.risk)
    sim.close()


@pytest.mark.parametrize("name,steps,mode,rank", [("cfg3_65536_moving", 10, 0, 0), ("cfg3_65536_moving", 3, 1, 0),
                                                  ("cfg2_4096_static", 10, 0, 0), ("cfg5_dense_720", 2, 0, 0),
                                                  ("cfg4_8x65536_moving", 6, 0, 5)],
                         ids=["cfg3-raster", "cfg3-allpairs", "cfg2", "cfg5", "cfg4-shard-of-rank-5"])
def test_full_size_batch_every_env_against_c_oracle(name, steps, mode, rank):
    """The benchmark configurations at their FULL sizes (65,536 vessels; 4,096
    vessels with 720 rays and 256 obstacles), every env and every buffer, bit
    for bit, against the C restatement of the oracle (itself pinned to the numpy
    text by tests/test_c_oracle_cpu.py) run on the host cores, after the reset
    and after every step.  The 8-GPU configuration is covered by the 65,536 envs
    that one of its ranks (rank 5) simulates: ranks share nothing while stepping."""
    import os

    from oracle import c_oracle

    cfg, scene = scenarios.make_workload(name, rank=rank)
    V = cfg["n_envs"]
    sim = make_sim(cfg, scene, lidar_mode=mode)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    env = c_oracle.COracleEnv(cfg, params.ray_table(cfg["n_rays"]), threads=threads)
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"], scene["poly_heading"])
    env.set_path(scene["path"])
    env.episode = np.zeros(V, dtype=np.int32)
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    assert_all_equal(sim, env, "full batch after reset")
    acts = scenarios.random_actions(V, steps, seed=43)
    for k in range(steps):
        sim.step_host(acts[k])
        env.step(acts[k])
        assert_all_equal(sim, env, f"full batch after step {k + 1}")
    env.close()
    sim.close()


SHAPES = [
    # V, R, C, P, NV, n_sides, tile, block
    (1, 1, 1, 1, 3, 3, 0, 0),
    (5, 7, 0, 2, 5, None, 0, 64),
    (67, 33, 3, 0, 5, None, 3, 128),
    (67, 180, 16, 5, 16, 16, 0, 0),
    (130, 1000, 2, 3, 7, 7, 1, 512),
    (130, 180, 5, 6, 5, None, 16, 256),
    (259, 90, 7, 1, 4, 4, 5, 0),
    (31, 360, 0, 9, 8, 6, 2, 512),
    (33, 720, 40, 40, 8, 8, 2, 1024),   # sixteen waves per workgroup (dense tiles take this size by default)
    (1000, 64, 1, 1, 3, 3, 64, 64),
    (9, 361, 3, 2, 4, 4, 0, 0),          # the work queue (360 rays and more) with a ray count that is not a multiple of four
    (21, 90, 70, 30, 3, 3, 2, 256),      # the work queue through four rounds of obstacles, 100 obstacles per env, 90 rays
    (1027, 180, 8, 8, 5, 5, 4, 256),   # the shape with a compile-time instantiation of the sensing kernel; last tile short
]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"V{s[0]}R{s[1]}C{s[2]}P{s[3]}NV{s[4]}t{s[6]}b{s[7]}" for s in SHAPES])
def test_odd_shapes_tiles_and_workgroup_sizes(shape):
    """Shapes off the benchmark grid (one ray, prime counts, no circles / no
    polygons, 16-vertex hulls with fewer sides in use, obstacle counts that are
    not powers of two, scans of 1,000 rays) under forced tile and workgroup
    sizes, with tight scenes so that vessels end up inside obstacles: every
    buffer against the C oracle, rasterising kernel, after reset and 6 steps."""
    from gym_auv_amd import _lib
    from oracle import c_oracle

    V, R, C, P, NV, n_sides, tile, block = shape
    cfg = params.make_config(V, C, P, NV, n_rays=R, max_steps=4)
    scene = scenarios.make_scene(V, C, P, NV, seed=V + R, extent=70.0, n_sides=n_sides, width=(6.0, 30.0))
    sim = make_sim(cfg, scene)
    if tile:
        sim.set_option(_lib.OPT_RASTER_TILE, tile)
    if block:
        sim.set_option(_lib.OPT_SENSE_BLOCK, block)
    env = c_oracle.COracleEnv(cfg, params.ray_table(R))
    env.set_scene(scene["circles"], scene["poly_pose"], scene["poly_verts"], scene["poly_nv"], scene["poly_heading"])
    env.set_path(scene["path"])
    env.episode = np.zeros(V, dtype=np.int32)
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    assert_all_equal(sim, env, "after reset")
    acts = scenarios.random_actions(V, 6, seed=R)
    acts[:, 0] = 1.0
    for k in range(6):
        sim.step_host(acts[k])
        env.step(acts[k])
        assert_all_equal(sim, env, f"after step {k + 1}")
    env.close()
    sim.close()


@MODES
def test_device_reproduces_the_committed_regression_vectors(mode):
    """tests/golden/sim_golden.npz: vectors made by the numpy oracle (not
    reference data; they pin the build's own arithmetic contract).  The device
    must produce them without any oracle in the loop."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sim_golden.npz"))
    scene = {k[len("scene_"):]: g[k] for k in g.files if k.startswith("scene_")}
    V = scene["path"].shape[0]
    cfg = params.make_config(V, scene["circles"].shape[1], scene["poly_pose"].shape[1], scene["poly_verts"].shape[2])
    sim = make_sim(cfg, scene, lidar_mode=mode)
    sim.reset(scene["init_state"])
    assert same_bits(sim.read("ranges"), g["reset_ranges"]) and same_bits(sim.read("hit"), g["reset_hit"])
    for k in range(g["actions"].shape[0]):
        sim.step_host(g["actions"][k])
        assert same_bits(sim.read("ranges"), g[f"ranges_{k}"]) and same_bits(sim.read("hit"), g[f"hit_{k}"])
        assert same_bits(sim.read("state"), g[f"state_{k}"])
    for name in PAIRS:
        if f"final_{name}" not in g.files:
            continue      # a buffer younger than the committed vectors
        want = g[f"final_{name}"]
        assert same_bits(sim.read(name), want.reshape(sim.read(name).shape)), name
    sim.close()
    # the same rollout with the closest-point-of-approach risk index
    sim = make_sim(dict(cfg, risk_model=0), scene, lidar_mode=mode)
    sim.reset(scene["init_state"])
    for k in range(g["actions"].shape[0]):
        sim.step_host(g["actions"][k])
    assert same_bits(sim.read("risk"), g["final_cpa_risk"]) and same_bits(sim.read("meta"), g["final_cpa_meta"])
    sim.close()


def test_device_action_pointer_and_dlpack_views():
    import torch

    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=130)
    sim = make_sim(cfg, scene)
    env = make_oracle(cfg, scene)
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    per = sim.tensor("perception")
    nav = sim.tensor("navigation")
    assert per.shape == (130, 1, 180) and per.dtype == torch.float32 and per.is_cuda
    assert nav.shape == (130, 1, 6) and nav.dtype == torch.float32
    assert per.data_ptr() == sim.buffer("perception")[0]          # zero-copy
    acts = scenarios.random_actions(130, 3, seed=51)
    st = torch.cuda.current_stream()
    for k in range(3):
        a = torch.from_numpy(acts[k]).cuda()
        sim.step_device(a.data_ptr(), stream=st)
        env.step(acts[k])
        st.synchronize()
        assert same_bits(per.cpu().numpy(), env.perception)        # view follows the step
        assert same_bits(nav.cpu().numpy(), env.navigation)
    assert_all_equal(sim, env, "device-pointer stepping")
    sim.close()


def test_vec_env_gym_contract():
    import torch

    from gym_auv_amd.env import VecAuvEnv

    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=64)
    venv = VecAuvEnv(cfg, scene, auto_reset=True)
    assert set(venv.observation_space.spaces) == {"perception", "navigation"}
    assert venv.observation_space.spaces["navigation"].shape[-1] == 6
    obs = venv.reset()
    assert obs["perception"].shape == (64, 1, 180) and obs["navigation"].shape == (64, 1, 6)
    # what radarCNN.py:190-193 does with the navigation observation
    assert obs["navigation"][:, 0, :].reshape(64, 6).shape == (64, 6)
    a = torch.rand(64, 2, device="cuda")
    a[:, 1] = a[:, 1] * 2 - 1
    obs, rew, done, info = venv.step(a)
    assert rew.shape == (64,) and done.shape == (64,) and done.dtype == torch.bool
    obs, rew, done, info = venv.step(a.cpu().numpy())
    assert torch.isfinite(obs["perception"]).all()
    venv.close()
    # restart on the device: the step after a done flag yields reward 0 and a cleared step counter
    cfg2 = dict(cfg, max_steps=3)
    venv = VecAuvEnv(cfg2, scene, auto_reset="device")
    venv.reset()
    for k in range(3):
        obs, rew, done, info = venv.step(a)
    assert done.all() and "terminal_mask" not in info        # truncated after 3 steps
    obs, rew, done, info = venv.step(a)
    assert (rew == 0).all() and (venv.sim.read("t_step") == 0).all()
    venv.close()
    # scenes from the device-side generator, a new one per episode
    venv = VecAuvEnv(cfg2, None, auto_reset="device", seed=11)
    venv.reset()
    first = venv.sim.read("circles").copy()
    for k in range(4):
        obs, rew, done, info = venv.step(a)
    assert (venv.sim.read("episode") == 1).all() and not np.array_equal(venv.sim.read("circles"), first)
    venv.close()


def test_facing_edge_rule_on_every_kind_of_polygon():
    """Convex counter-clockwise polygons show the sensor only their facing edges once the vessel is beyond their
    farthest vertex; clockwise, concave, degenerate and too-short vertex lists never do.  Vessels on a grid that
    covers inside, on the reach circle and far outside; both device paths against the numpy and the C oracle."""
    from oracle import c_oracle

    side = 21
    V = side * side
    P, NV = 6, 6
    cfg = params.make_config(V, 0, P, NV)
    hexagon = scenarios.ngon(np.full(6, 12.0))
    shapes = [
        (hexagon, 6),                                                         # convex, counter-clockwise
        (hexagon[::-1].copy(), 6),                                            # the same, clockwise
        (np.array([[-10, -6], [10, 0], [-10, 6], [-4, 0], [0, 0], [0, 0.0]]), 4),   # concave arrow
        (np.array([[-8, 0], [0, 0], [8, 0], [0, 0], [0, 0], [0, 0.0]]), 3),         # three collinear points
        (scenarios.ship_polygon(9.0).reshape(5, 2).tolist() + [[0.0, 0.0]], 5),     # a ship
        (np.array([[-5, -5], [5, -5], [5, 5], [-5, 5], [0, 0], [0, 0.0]]), 4),      # a square
    ]
    verts = np.zeros((V, P, NV, 2))
    nv = np.zeros((V, P), dtype=np.int32)
    pose = np.zeros((V, P, 6))
    heading = np.zeros((V, P))
    centres = [(-40.0, -25.0), (0.0, -25.0), (40.0, -25.0), (-40.0, 25.0), (0.0, 25.0), (40.0, 25.0)]
    for q, ((sh, n), (cx, cy)) in enumerate(zip(shapes, centres)):
        verts[:, q] = np.asarray(sh, dtype=np.float64)
        nv[:, q] = n
        h = 0.3 * q
        pose[:, q] = (cx, cy, np.cos(h), np.sin(h), 0.2, -0.1)
        heading[:, q] = h
    gx, gy = np.meshgrid(np.linspace(-60, 60, side), np.linspace(-45, 45, side))
    init = np.zeros((6, V))
    init[0], init[1], init[2] = gx.ravel(), gy.ravel(), np.linspace(-3, 3, V)
    scene = {"circles": np.zeros((V, 0, 5)), "poly_pose": pose, "poly_verts": verts, "poly_nv": nv,
             "poly_heading": heading, "init_state": init, "path": scenarios._straight_path(V, init[2])}
    r2 = O.polygon_reach2(verts, nv)[0]
    assert (r2 >= 0).tolist() == [True, False, False, False, True, True]
    env = make_oracle(cfg, scene)
    cenv = c_oracle.COracleEnv(cfg, params.ray_table(cfg["n_rays"]), threads=2)
    cenv.set_scene(scene["circles"], pose, verts, nv, heading)
    cenv.set_path(scene["path"])
    env.reset(init)
    cenv.reset(init)
    assert same_bits(cenv.ranges, env.ranges) and same_bits(cenv.hit, env.hit)
    acts = scenarios.random_actions(V, 3, seed=12)
    for mode in (0, 1):
        sim = make_sim(cfg, scene, lidar_mode=mode)
        sim.reset(init)
        env.reset(init)
        assert_all_equal(sim, env, f"facing-edge rule, mode {mode}, reset")
        for k in range(3):
            sim.step_host(acts[k])
            env.step(acts[k])
            assert_all_equal(sim, env, f"facing-edge rule, mode {mode}, step {k}")
        sim.close()
    cenv.close()
    # the rule changes nothing here that a full test of every edge would see
    full = O.lidar(env.state, env.circles, env.poly_pose, env.poly_verts, env.poly_nv, env.ray_dirs,
                   cfg["sensor_range"], np.full((V, P), -1.0))
    assert same_bits(full[0], env.ranges) and same_bits(full[1], env.hit)
    assert env.collision.any() and not env.collision.all()


@pytest.mark.parametrize("mode", [0, 1], ids=["raster", "allpairs"])
def test_polyline_paths_match_the_oracle(mode):
    """Piecewise-straight paths (up to 4 segments, a third of the envs with fewer): navigation features,
    along-track position, reward, goal and the segment each vessel is on, bit for bit, while the vessels steer
    towards their look-ahead points round the corners; with the in-step restart (rasterising path) a
    restarting vessel goes back to the first segment."""
    from gym_auv_amd import _lib
    from tests.helpers import winding_paths

    V = 333
    cfg, scene = scenarios.make_workload("cfg3_65536_moving", n_envs=V)
    cfg["n_path_segments"] = 4
    seg, nseg, psi0 = winding_paths(V, K=5, seed=8)
    scene["path"], scene["path_nseg"] = seg, nseg
    scene["init_state"][2] = psi0
    sim = make_sim(cfg, scene, lidar_mode=mode)
    env = make_oracle(cfg, scene)
    if mode == 0:
        sim.set_option(_lib.OPT_AUTO_RESET, 1)
        env.auto_reset = True
    sim.reset(scene["init_state"])
    env.reset(scene["init_state"])
    assert_all_equal(sim, env, "polyline paths after reset")
    turned = np.zeros(V, dtype=bool)
    restarted = False
    for k in range(150):
        act = np.stack([np.full(V, 1.0), np.clip(2.0 * env.navigation[:, 0, 3].astype(np.float64), -1, 1)])
        sim.step_host(act)
        env.step(act)
        if k % 7 == 0 or k > 140:
            assert_all_equal(sim, env, f"polyline paths, step {k}")
        turned |= env.path_seg > 0
        restarted |= bool((env.episode > 0).any())
    assert_all_equal(sim, env, "polyline paths, end")
    assert turned.mean() > 0.5 and (mode == 1 or restarted)
    # more segments than the batch has room for
    with pytest.raises(_lib.AuvError, match="room for"):
        sim.set_path(np.zeros((V, 5, 6)))
    sim.close()
