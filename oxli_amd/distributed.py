"""Multi-GPU shard merge for ``KmerCountTable`` -- one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference defines how two shards combine: ``add`` (lib.rs:778-837) -- per-key sum of counts,
sum of ``consumed``.  Records are independent (one ``consume`` per record, README.md:96-98), so
each rank counts its own records into its own device table with no communication, and ONE
exchange at the end makes the result global:

    owner(hash) = floor(hi32(hash) * world / 2^32)          (a contiguous slice of hash space)
    every rank sends each owner the (hash, count) pairs it holds for that owner's slice
    (all-to-all: all 7 xGMI links of a GPU carry traffic at once), and each owner folds what it
    receives into a fresh table.  Afterwards rank r holds exactly the keys of slice r with their
    global counts; the global table is the disjoint union over ranks.

An element-wise all-reduce of the raw tables would be wrong: open addressing places a key
wherever its probe sequence found room, so slot layouts differ between ranks.

Only plumbing lives here (bucketing with torch ops, the collective); hashing and counting stay
in the HIP library.
"""
import torch
import torch.distributed as dist

__all__ = ["owner_of", "partition_by_owner", "exchange_route", "exchange_pairs", "merge_across_ranks", "global_scalar_sum",
           "consume_device_early", "EARLY_MODES"]


def owner_of(hashes_i64: torch.Tensor, world: int) -> torch.Tensor:
    """Owner rank of each hash (hashes carried as int64 bit patterns)."""
    hi32 = (hashes_i64 >> 32) & 0xFFFFFFFF
    return (hi32 * world) >> 32


def partition_by_owner(hashes_i64, counts_i64, world):
    """Buckets the pairs by owner.  Returns (pairs [n, 2] with owner p's rows contiguous, send_counts[world])."""
    if hashes_i64.numel() == 0:
        return torch.empty((0, 2), dtype=torch.int64, device=hashes_i64.device), torch.zeros(world, dtype=torch.int64)
    own = owner_of(hashes_i64, world)
    order = torch.argsort(own, stable=True)
    send_counts = torch.bincount(own, minlength=world).to(torch.int64).cpu()
    return torch.stack([hashes_i64[order], counts_i64[order]], dim=1).contiguous(), send_counts


import weakref

_ROUTES = weakref.WeakKeyDictionary()  # process group object -> route (a destroyed group takes its entry along)


def exchange_route(group=None):
    """Which collective moves the pairs: "all_to_all" (RCCL / gloo ``all_to_all_single`` with uneven splits) or
    "all_gather" (every rank publishes its whole bucketed list; world x the traffic, for back ends without an uneven
    all-to-all).  Chosen up front -- ``KCT_A2A_FALLBACK=1`` asks for the second -- and agreed between the ranks with one
    tiny all-reduce, so that every rank issues the same collectives; an error in a collective is never caught (the
    communicator is unusable after one anyway, and a rank that switched routes alone would hang the others)."""
    import os
    key = group if group is not None else dist.group.WORLD  # (the default group's object changes with every init_process_group)
    if key not in _ROUTES:  # agreed once per process group
        mine = 1 if os.environ.get("KCT_A2A_FALLBACK") == "1" else 0
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.tensor([mine], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        _ROUTES[key] = "all_gather" if int(t.item()) else "all_to_all"
    return _ROUTES[key]


def exchange_pairs(pairs, send_counts, zero_count=0, group=None, route=None):
    """All-to-all of owner-bucketed pairs.  ``pairs`` is an int64 tensor [n, 2] = (hash, count),
    owner p's rows contiguous and in rank order.  Returns (recv_pairs [m, 2], zero_total).

    ``zero_count`` is this rank's count for hash 0, which the library keeps outside the device
    table (0 is its EMPTY sentinel); it rides along with the size exchange to its owner, rank 0.
    """
    world = dist.get_world_size(group)
    assert world == send_counts.numel()
    dev = pairs.device
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        # debugging aid (several ranks sharing one GPU, no RCCL): stage the collective through host memory
        out, zero_total = exchange_pairs(pairs.cpu(), send_counts, zero_count, group, route)
        return out.to(dev), zero_total
    rank = dist.get_rank(group)
    if route is None:
        route = exchange_route(group)
    if route == "all_to_all":
        # 1) how much will I receive from each peer (tiny fixed-layout exchange)
        meta = torch.zeros((world, 2), dtype=torch.int64)
        meta[:, 0] = send_counts
        meta[0, 1] = int(zero_count)
        meta = meta.to(dev)
        got = torch.empty_like(meta)
        dist.all_to_all_single(got, meta, group=group)
        got = got.cpu()
        recv_counts = got[:, 0]
        zero_total = int(got[:, 1].sum())
        # 2) the pairs themselves: one collective moves hashes and counts together
        out = torch.empty((int(recv_counts.sum()), 2), dtype=torch.int64, device=dev)
        dist.all_to_all_single(out, pairs, output_split_sizes=recv_counts.tolist(), input_split_sizes=send_counts.tolist(),
                               group=group)
        return out, zero_total
    # all_gather route: every rank publishes its whole bucketed list (padded to the longest) and its bucket sizes; rank r
    # keeps bucket r of everybody.
    meta = torch.zeros(world + 2, dtype=torch.int64)
    meta[:world] = send_counts
    meta[world] = int(zero_count)
    meta[world + 1] = pairs.shape[0]
    metas = [torch.empty_like(meta).to(dev) for _ in range(world)]
    dist.all_gather(metas, meta.to(dev), group=group)
    metas = torch.stack(metas).cpu()
    longest = int(metas[:, world + 1].max())
    padded = torch.zeros((max(longest, 1), 2), dtype=torch.int64, device=dev)
    padded[: pairs.shape[0]] = pairs
    lists = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(lists, padded, group=group)
    parts = []
    for src in range(world):
        counts = metas[src, :world]
        start = int(counts[:rank].sum())
        parts.append(lists[src][start:start + int(counts[rank])])
    out = torch.cat(parts) if parts else torch.empty((0, 2), dtype=torch.int64, device=dev)
    return out, (int(metas[:, world].sum()) if rank == 0 else 0)  # key 0's owner is rank 0


def global_scalar_sum(value: int, device, group=None) -> int:
    if dist.get_backend(group) == "gloo":
        device = "cpu"
    t = torch.tensor([value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def merge_across_ranks(table, group=None):
    """Turns per-rank tables into the owner-partitioned global table, in place.

    After the call ``table`` on rank r holds every key of hash-slice r with its global count;
    ``len`` / ``sum_counts`` / ``consumed`` of the global table are the sums over ranks
    (``global_scalar_sum``).  ``table.consumed`` keeps this rank's own share.
    Returns the number of pairs this rank received.

    Device work is native: ``kct_export_by_owner_device`` buckets the table by owner in two
    kernels, ``kct_merge_pairs_device`` folds what arrives.  ``partition_by_owner`` above is the
    same bucketing in torch ops (what the CPU gloo test exercises).

    After the exchange the table is RESIZED for what this rank owns (``kct_resize``): while counting, a rank's private
    table has to hold every k-mer its reads touch (close to the whole genome), but an owner's table holds 1/world of
    the key space (2^27 slots per GPU for C4 instead of 2^30).

    A ``store_kmers`` table is refused: its hash -> k-mer map lives on the host of the rank that saw the k-mer, and this
    exchange moves (hash, count) pairs only.
    """
    import ctypes as C

    import numpy as np

    world = dist.get_world_size(group)
    if world == 1:
        return 0
    if getattr(table, "store_kmers", False):
        raise ValueError("merge_across_ranks moves (hash, count) pairs only: a store_kmers table would lose its hash -> k-mer map")
    dev = torch.device("cuda", torch.cuda.current_device())
    n = len(table)
    pairs = torch.empty((max(n, 1), 2), dtype=torch.int64, device=dev)
    part_counts = np.zeros(world, dtype=np.uint64)
    got = C.c_uint64()
    table._check(table._lib.kct_export_by_owner_device(table._h, world, C.c_void_p(pairs.data_ptr()), n, part_counts.ctypes.data,
                                                       C.byref(got)))
    send_counts = torch.from_numpy(part_counts.astype(np.int64))
    recv, zero = exchange_pairs(pairs[: got.value], send_counts, table.get_hash(0), group)
    torch.cuda.synchronize()
    consumed = table.consumed
    table.clear()
    table.resize(recv.shape[0])  # an owner's table: sized for its slice of the key space
    if zero:
        zk, zc = np.zeros(1, dtype=np.uint64), np.array([zero], dtype=np.uint64)
        table._check(table._lib.kct_merge_host(table._h, zk.ctypes.data, zc.ctypes.data, 1, None, None))
    a, b = C.c_uint64(), C.c_uint64()
    table._check(table._lib.kct_merge_pairs_device(table._h, C.c_void_p(recv.data_ptr()), recv.shape[0], C.byref(a), C.byref(b)))
    table._check(table._lib.kct_add_consumed(table._h, consumed))
    return recv.shape[0]


# ---- the EARLY route: entries travel to their owner while they are counted (csrc/kct_route.hip) -----------------------------
EARLY_MODES = {"hash": 0, "dedupe64": 1, "compact": 2}


class _Exchanger:
    """The two callbacks ``kct_consume_device_routed`` needs, over ``torch.distributed``: device buffers come from torch
    (so that the collective can take them as tensors), the all-to-all is ``all_to_all_single`` on bytes with uneven splits
    (RCCL over xGMI with the nccl backend; staged through host memory when ranks share a GPU under gloo)."""

    def __init__(self, group, dev):
        import ctypes as C
        self.group, self.dev = group, dev
        self.world = dist.get_world_size(group)
        self.host = dist.get_backend(group) != "nccl"
        self.keep = {}      # device address -> tensor (send buffers handed to the library)
        self.recv = []      # received buffers, alive until the routed call returns
        self.error = None
        self.bytes_sent = 0
        self._alloc = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)(self.alloc)
        self._xchg = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_uint64))(self.exchange)
        self.alloc_ptr = C.cast(self._alloc, C.c_void_p)
        self.xchg_ptr = C.cast(self._xchg, C.c_void_p)

    def alloc(self, _user, nbytes):
        try:
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.dev)
            self.keep[t.data_ptr()] = t
            return t.data_ptr()
        except Exception as e:  # noqa: BLE001 -- reported through the library's status
            self.error = e
            return None

    def exchange(self, _user, d_send, send_elems, elem_bytes, d_recv, recv_elems):
        try:
            world, eb = self.world, int(elem_bytes)
            send = [int(send_elems[i]) for i in range(world)]
            buf = self.keep[int(d_send)][: sum(send) * eb]
            meta = torch.tensor(send, dtype=torch.int64, device="cpu" if self.host else self.dev)
            got = torch.empty_like(meta)
            dist.all_to_all_single(got, meta, group=self.group)
            recv = [int(v) for v in got.cpu().tolist()]
            src = buf.cpu() if self.host else buf
            out = torch.empty(sum(recv) * eb, dtype=torch.uint8, device=src.device)
            dist.all_to_all_single(out, src, output_split_sizes=[r * eb for r in recv], input_split_sizes=[s_ * eb for s_ in send],
                                   group=self.group)
            if self.host:
                out = out.to(self.dev)
            torch.cuda.synchronize()
            if out.numel() == 0:
                out = torch.empty(256, dtype=torch.uint8, device=self.dev)
            self.recv.append(out)
            self.bytes_sent += (sum(send) - send[dist.get_rank(self.group)]) * eb
            d_recv[0] = out.data_ptr()
            for i in range(world):
                recv_elems[i] = recv[i]
            return 0
        except Exception as e:  # noqa: BLE001
            self.error = e
            return 1


def consume_device_early(table, data_ptr, nbytes, consumed_bytes, group=None, mode="auto", max_windows=None):
    """Counts this rank's device-resident record stream by the EARLY route: K1 here, entries to their owner GPUs with three
    all-to-alls, K1b / K2 on the owners (``kct_consume_device_routed``).  Every rank must call it, with tables of one
    capacity.  ``mode``: "compact" (k <= 21) and "dedupe64" (k <= 32) count packed k-mers first and hash each distinct one
    when the table is read -- right for deep coverage, where an owner meets every k-mer many times; "hash" hashes every window
    (any k <= 64; low coverage); "auto" picks by k.  Returns (k-mers this rank counted as an owner, stats dict).

    A pass keeps K1's regions, the send and receive buffers and K1b's regions in HBM at once (~4.4 entries of 4 or 8 bytes per
    window start): a stream too long for that is cut into passes of ``max_windows`` window starts (default: what 60 % of the
    free HBM of the tightest rank allows), each with its own exchange; consecutive passes overlap by k - 1 bytes, so no window
    is lost or counted twice.

    Afterwards the ranks' tables are a disjoint partition of the key space (by k-mer slice for the dedupe-first modes, by
    hash slice -- the late route's owner rule -- for "hash"): ``global_scalar_sum`` of ``len`` / ``sum_counts`` gives the
    global table's, and no ``merge_across_ranks`` is needed."""
    import ctypes as C

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if getattr(table, "store_kmers", False):
        raise ValueError("the early route moves packed k-mers / hashes only: a store_kmers table would lose its hash -> k-mer map")
    if mode == "auto":
        mode = "compact" if table.ksize <= 21 else "dedupe64" if table.ksize <= 32 else "hash"
    k = table.ksize
    dev = torch.device("cuda", torch.cuda.current_device())
    host = dist.get_backend(group) != "nccl"
    cdev = "cpu" if host else dev
    # one capacity on every rank (the senders' bins follow the owners' table geometry); the passes every rank will make
    esz = 4 if mode == "compact" else 8
    windows = max(int(nbytes) - k + 1, 0)
    if max_windows is None:
        free = torch.cuda.mem_get_info()[0]
        max_windows = max(1 << 24, int(0.6 * free / (4.4 * esz)))
    max_windows = max(1 << 16, int(max_windows) & ~0xFFFF)
    mine = torch.tensor([table.capacity, -table.capacity, -(-windows // max_windows) if windows else 1], dtype=torch.int64, device=cdev)
    dist.all_reduce(mine, op=dist.ReduceOp.MAX, group=group)
    if int(mine[0]) != -int(mine[1]):
        raise ValueError(f"the early route needs tables of one capacity on every rank (have {-int(mine[1])} .. {int(mine[0])} slots)")
    passes = max(1, int(mine[2]))
    step = ((-(-windows // passes)) + 0xFFFF) & ~0xFFFF if windows else 0   # window starts per pass, a multiple of 2^16 (16-byte aligned cuts)
    keys = ("entries_sent", "entries_received", "entry_bytes", "overflow_sent", "overflow_received", "exchange_us", "blocks_abandoned", "skewed")
    total_n, agg, bytes_sent, err, status = 0, dict.fromkeys(keys, 0), 0, None, 0
    for p in range(passes):
        off = min(p * step, int(nbytes))
        length = min(int(nbytes) - off, step + k - 1) if p + 1 < passes else int(nbytes) - off
        ex = _Exchanger(group, dev)
        n, stats = C.c_uint64(), (C.c_uint64 * 8)()
        st = table._lib.kct_consume_device_routed(table._h, C.c_void_p(int(data_ptr) + off), max(length, 0), int(consumed_bytes) if p == 0 else 0,
                                                  world, rank, EARLY_MODES[mode], ex.alloc_ptr, ex.xchg_ptr, None, C.byref(n), stats)
        total_n += n.value
        for kk, v in zip(keys, stats):
            agg[kk] = int(v) if kk in ("entry_bytes",) else agg[kk] + int(v)
        bytes_sent += ex.bytes_sent
        err = err or ex.error
        status = status or st
        del ex
    # a failure on ANY rank is every rank's failure (tables are then inconsistent across the job)
    bad = torch.tensor([1 if status != 0 else 0], dtype=torch.int64, device=cdev)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    if err is not None:
        raise err
    table._check(status)
    if int(bad.item()):
        raise RuntimeError("the early route failed on another rank: clear the tables and use the late route (merge_across_ranks)")
    return total_n, dict(agg, mode=mode, bytes_sent=bytes_sent, passes=passes)
