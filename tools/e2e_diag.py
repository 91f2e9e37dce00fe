#!/usr/bin/env python3
"""Round 6: where does a kct_consume_batch call of C2's 151 MB batch spend its time?  One process per setting (the packer pool and the
pinning are fixed at first use): KCT_PACK_THREADS x KCT_PACK_PIN, three processes each; every process prints the timeline
(kct_batch_timeline) of its median call of seven.

    python tools/e2e_diag.py            # the matrix (spawns children)
    python tools/e2e_diag.py child      # one process: one line of JSON
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import torch
    from oxli_amd import KmerCountTable, _lib
    lib = _lib.load()
    G, R, L, k = 5_000_000, 1_000_000, 150, 21
    g = torch.empty(G, dtype=torch.uint8, device="cuda")
    r = torch.empty(R * (L + 1), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    lib.kct_synth_genome_device(g.data_ptr(), G, 42, s)
    lib.kct_synth_reads_device(r.data_ptr(), g.data_ptr(), G, 0, R, L, 1337, s)
    torch.cuda.synchronize()
    host = r.cpu().numpy().reshape(R, L + 1)
    flat = np.ascontiguousarray(host[:, :L]).reshape(-1)
    offsets = np.arange(R + 1, dtype=np.uint64) * np.uint64(L)
    t = KmerCountTable(k, capacity=G)
    runs = []
    for i in range(9):
        t.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = t.consume_batch((flat, offsets))
        t1 = time.perf_counter()
        t.sync()
        t2 = time.perf_counter()
        assert n == R * (L - k + 1)
        if i >= 2:
            runs.append((t2 - t0, t1 - t0, t.batch_timeline()))
    runs.sort(key=lambda x: x[0])
    total, call, tl = runs[len(runs) // 2]
    out = {"threads_env": os.environ.get("KCT_PACK_THREADS", "16"), "pin": os.environ.get("KCT_PACK_PIN", "0"), "kmers_per_s": round(R * (L - k + 1) / total),
           "total_ms": round(total * 1e3, 3), "call_ms": round(call * 1e3, 3), "min_max_ms": [round(runs[0][0] * 1e3, 3), round(runs[-1][0] * 1e3, 3)],
           "timeline": {kk: round(v, 3) for kk, v in tl.items()}}
    out["pack_GB_per_s_wall"] = round(tl["source_bytes"] / max(1e-9, (tl["last_packer_end_ms"] - tl["first_packer_start_ms"]) * 1e-3) / 1e9, 1)
    out["pack_GB_per_s_per_thread"] = round(tl["source_bytes"] / max(1e-9, tl["threads_busy_ms_sum"] * 1e-3) / 1e9, 2)
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        return
    for threads in ("16", "32"):
        for pin in ("0", "1"):
            for rep in range(3):
                env = dict(os.environ, KCT_PACK_THREADS=threads, KCT_PACK_PIN=pin)
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
                line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                print(line[-1] if line else f"threads={threads} pin={pin}: FAILED {p.stderr[-400:]}", flush=True)


if __name__ == "__main__":
    main()
