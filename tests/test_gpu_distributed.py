"""The multi-GPU merge with the DEVICE tables under real multi-process conditions (``-m gpu``): 2 and 4 ranks share the one
GPU of the test box over gloo (the pair exchange is staged through host memory; on an 8-GPU node the same code runs over
RCCL).  Each rank counts its shard on the device, exports its table bucketed by owner with the native kernels, exchanges,
merges what it receives into an owner-sized table; rank 0 compares the union of the owner tables with the oracle."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,k,per_rank,genome", [(2, 21, 40_000, 300_000), (4, 31, 20_000, 200_000), (3, 21, 400_000, 2_000_000)])
def test_ranks_sharing_one_gpu_merge_to_the_oracle_table(world, k, per_rank, genome):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(k), str(per_rank), str(genome)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-6000:]
    assert f"DIST_GPU_OK world={world}" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("world,k,per_rank,genome,mode", [(2, 21, 400_000, 2_000_000, "compact"), (3, 21, 300_000, 2_000_000, "compact"),
                                                           (4, 31, 200_000, 2_000_000, "dedupe64"), (2, 51, 200_000, 2_000_000, "hash"),
                                                           (3, 21, 300_000, 4_000_000, "hash"), (4, 21, 800_000, 6_000_000, "compact")])
def test_ranks_sharing_one_gpu_early_route_equals_the_oracle_table(world, k, per_rank, genome, mode):
    """The EARLY route (kct_consume_device_routed): K1 on every rank with owner-grouped bins, three all-to-alls
    (gloo here, staged through the host), K1b / K2 on the owners -- C4-shaped input (150 bp reads, deep coverage), two passes
    per rank.  The union of the owners' tables must be the oracle's table of the whole stream."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(k), str(per_rank), str(genome), f"early:{mode}"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-6000:]
    assert f"DIST_GPU_OK world={world}" in out.stdout and f"route=early:{mode}" in out.stdout, out.stdout[-2000:]
