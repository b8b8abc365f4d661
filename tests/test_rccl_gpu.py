"""The collectives of the multi-GPU path through real RCCL.  One GPU allows one rank (RCCL refuses two ranks on a
device), so the group has a single member: what runs is RCCL's initialisation on the device, its barrier,
all-reduce and gather kernels and its teardown, from the code the 8-GPU run uses; each in a child process, so that
the test runner's own GPU state is untouched."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_every_collective_of_the_dist_layer_through_rccl():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), str(_free_port())],
                       env=_clean_env(), capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["stats_equal"] and out["on_device_equal"] and out["env_steps"] == 512.0
    assert out["all_reduce_max"] == [1.5, -2.0]
    assert out["gather_into_out"] and out["gather_new"] and out["gather_samples"] and out["destroyed"]


def test_bench_line_of_a_grouped_run_through_rccl():
    """bench.py with its process group forced on the one rank: the multi-GPU code path (group creation on the
    device, barrier-bracketed timing, statistics all-reduce inside the timed region, scan gather on the trainer
    rank) on the "nccl" backend, and the fields its line then carries."""
    env = _clean_env()
    env.update(AUV_BENCH_FORCE_GROUP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2",
                        "--repeats", "2", "--envs", "8192", "--spin-seconds", "0.2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["ranks_joined"] == 1 and out["value"] > 0
    c = out["collectives"]
    assert c["backend"] == "nccl" and c["stats_allreduce_ms"] > 0
    assert c["scan_gather"]["checked"] is True and c["scan_gather"]["ms"] > 0
    assert c["scan_gather"]["bytes_per_rank"] == 8192 * 180 * 4
    assert out["episode_stats"]["env_steps"] == 8192.0
