"""Randomised cross-check of the four device paths (tools/stress_paths.py): random k <= 32, genome size, read length,
coverage, error / N rates and call patterns; every path must build the same table."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_paths_agree_on_random_workloads(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_paths.py"), "25", str(seed)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all paths agree" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
