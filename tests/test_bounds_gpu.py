"""The library built with its on-chip index-bound checks (-DAUV_DEBUG_BOUNDS,
gym-auv_amd/libauvsim_debug.so) run over crowded scenes, overflowing pair lists
and every tile / workgroup shape: no bound may be violated.  Parity tests cannot
see a stray LDS write that lands in unused bytes; this can."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_index_bound_is_violated():
    dbg = os.path.join(ROOT, "gym-auv_amd", "libauvsim_debug.so")
    if not os.path.exists(dbg):
        pytest.fail(f"{dbg} is missing: run __graft_entry__.build()")
    env = dict(os.environ, AUVSIM_LIB=dbg, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bounds_driver.py")], capture_output=True,
                       text=True, env=env, timeout=280)
    assert r.returncode == 0 and "violations 0 rc 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
