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


@pytest.mark.parametrize("world,k,per_rank,genome", [(2, 21, 40_000, 300_000), (4, 31, 20_000, 200_000), (3, 21, 400_000, 2_000_000), (8, 21, 150_000, 2_000_000)])
def test_ranks_sharing_one_gpu_merge_to_the_oracle_table(world, k, per_rank, genome):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(k), str(per_rank), str(genome)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-6000:]
    assert f"DIST_GPU_OK world={world}" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("world,k,per_rank,genome,path,L", [
    (2, 21, 400_000, 2_000_000, "auto", 150), (3, 21, 300_000, 2_000_000, "dedupe", 150), (4, 31, 200_000, 2_000_000, "dedupe", 150),
    (2, 51, 200_000, 2_000_000, "auto", 150), (3, 21, 300_000, 4_000_000, "partitioned", 150), (4, 21, 800_000, 6_000_000, "auto", 150),
    # the target world: 8 ranks (7 peers each, owners of an eighth of the minimisers), C4-shaped
    (8, 21, 300_000, 6_000_000, "auto", 150), (8, 31, 150_000, 3_000_000, "partitioned", 150),
    # C5-shaped: 10 kbp reads at k = 51, passes of 2^22 window starts cut inside records
    (8, 51, 2_400, 3_000_000, "auto", 10_000), (2, 51, 6_000, 2_000_000, "partitioned", 10_000)])
def test_ranks_sharing_one_gpu_early_route_equals_the_oracle_table(world, k, per_rank, genome, path, L):
    """The EARLY route (kct_consume_device_routed): every rank cuts its records into super-k-mers by owner, the parts
    cross in pipelined all-to-alls (gloo here, staged through the host), every owner counts what it receives with the table's ordinary
    bulk path.  The union of the owners' tables must be the oracle's table of the whole stream (add() semantics, lib.rs:778-837)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(k), str(per_rank), str(genome), f"early:{path}", str(L)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-6000:]
    assert f"DIST_GPU_OK world={world}" in out.stdout and f"route=early:{path}" in out.stdout, out.stdout[-2000:]


def test_a_failing_rank_ends_the_early_route_on_every_rank():
    """Fault injection (kct_debug_inject_fault): 4 ranks sharing the GPU over gloo, ONE rank fails at one point of the early route's
    protocol -- seven scenarios, among them rank 2 at the split of pass 1 -- and every rank must return an error from that call within
    30 s with its slabs released; a clean job through the same processes then equals the oracle (dist_gpu_worker.py::faults)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "21", "150000", "2000000", "fault", "150"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-6000:]
    assert "FAULTS_OK world=4 scenarios=7" in out.stdout and "DIST_GPU_OK world=4" in out.stdout, out.stdout[-2000:]


def test_store_kmers_tables_merge_with_their_k_mer_maps():
    """The late route with store_kmers tables (lib.rs:810-828: add() merges hash_to_kmer): after merge_across_ranks every rank can
    unhash exactly the keys it owns, with the global counts."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "21", "300", "20000", "late:store_kmers"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-6000:]
    assert "DIST_GPU_OK world=2" in out.stdout and "route=late:store_kmers" in out.stdout, out.stdout[-2000:]


def test_one_rank_through_a_real_rccl_communicator():
    """torch.distributed's nccl back end IS RCCL on this box: a group of one rank runs the exchanges' device-side, asynchronous branches
    (what 8 ranks run over xGMI) through a real communicator -- the early route's size / payload all-to-alls under pipelined passes and
    the late route's pair exchange; table == oracle."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "21", "200000", "2000000", "rccl-alone"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-6000:]
    assert "DIST_GPU_OK world=1" in out.stdout and "route=rccl-alone" in out.stdout and "native=ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def _gpus():
    import torch
    return torch.cuda.device_count()   # (counting devices does not initialise the GPU)


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
@pytest.mark.parametrize("native", ["0", "1"], ids=["torch.distributed", "libkct_rccl.so"])
@pytest.mark.parametrize("route", ["late", "early:auto"])
def test_two_gpus_over_rccl(route, native):
    """Both routes between TWO GPUs over RCCL / xGMI -- the cross-rank stream ordering, the self send / receive inside a group and the
    error agreement that a world of one cannot show -- through torch.distributed's nccl back end and through libkct_rccl.so's own
    communicator (NativeRccl); union of the owners' tables == the oracle's table.  Skipped on the one-GPU test boxes of this pool."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", KCT_DIST_BACKEND="nccl", KCT_DIST_NATIVE=native)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), "21", "400000", "2000000", route]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + "\n" + out.stderr[-6000:]
    assert "DIST_GPU_OK world=2" in out.stdout, out.stdout[-2000:]
