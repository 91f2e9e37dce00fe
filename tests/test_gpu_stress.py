"""Randomised cross-checks (tools/stress_paths.py, tools/stress_merge.py): random k, genome size, read length, coverage, error /
N rates and call patterns -- every device path (direct, partitioned, dedupe-first forced and automatic, packed input, the early
route's loop-back) must build the same table; add() and export -> merge must equal consuming both read sets."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_paths_agree_on_random_workloads(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_paths.py"), "25", str(seed)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all paths agree" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("seed", [21, 22])
def test_merge_routes_agree_on_random_workloads(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_merge.py"), "8", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "all merges agree" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
