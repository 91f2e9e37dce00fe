"""The early route's kernels under rocprofv3 (loop-back exchange, one GPU): C2-sized input, the three modes, two passes each."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oxli_amd import KmerCountTable, _lib
lib = _lib.load()
G, R, L = 5_000_000, 1_000_000, 150
g = torch.empty(G, dtype=torch.uint8, device="cuda"); r = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
lib.kct_synth_genome_device(g.data_ptr(), G, 42, s); lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, R, L, 1337, s); torch.cuda.synchronize()
for k, mode in ((21, 2), (31, 1), (51, 0)):
    t = KmerCountTable(k, capacity=G)
    n, st = C.c_uint64(), (C.c_uint64 * 8)()
    for _ in range(2):
        t._check(lib.kct_consume_device_routed(t._h, C.c_void_p(r.data_ptr()), r.numel(), R * L, 1, 0, mode, None, None, None, C.byref(n), st))
    assert t.sum_counts == 2 * R * (L - k + 1)
print("ok")
