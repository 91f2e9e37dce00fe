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
           "consume_device_early", "ROUTE_STATS", "NativeRccl"]


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


class NativeRccl:
    """A communicator of libkct_rccl.so (include/kct_rccl.h) for this process: what a Rust or C caller of the C ABI uses -- the early
    route's ``kct_exchange_ops`` and the late route's ``kct_rccl_merge_across_ranks``, ncclSend / ncclRecv groups on a stream of the
    helper's own, no Python between the collectives.  The 128-byte ``ncclUniqueId`` is made on rank 0 and handed round through the
    ``torch.distributed`` group that is already up (any back end: it is 128 bytes, once).  One rank per GPU (RCCL refuses two ranks on
    one device); in a process that carries PyTorch the helper binds to PyTorch's own librccl.so.1 -- one copy of RCCL per process."""

    def __init__(self, group=None, device=None):
        import ctypes as C

        import numpy as np

        from . import _lib
        self._lib = _lib.load_rccl()
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        dev = torch.cuda.current_device() if device is None else int(device)
        ident = np.zeros(_lib.RCCL_ID_BYTES, dtype=np.uint8)
        if rank == 0 and self._lib.kct_rccl_unique_id(ident.ctypes.data) != 0:
            raise RuntimeError(self.last_error())
        if world > 1:
            box = [ident.tobytes()]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = np.frombuffer(box[0], dtype=np.uint8).copy()
        h = C.c_void_p()
        if self._lib.kct_rccl_create(ident.ctypes.data, world, rank, dev, C.byref(h)) != 0:
            raise RuntimeError(self.last_error())
        self._h, self.world, self.rank = h, world, rank
        self.ops = C.c_void_p(self._lib.kct_rccl_ops(h))

    def last_error(self):
        return (self._lib.kct_rccl_last_error() or b"").decode("utf-8", "replace")

    def merge_when_alone(self, on=True):
        self._lib.kct_rccl_merge_when_alone(self._h, 1 if on else 0)

    def merge_across_ranks(self, table):
        """The late route's collective, natively (``kct_rccl_merge_across_ranks``).  Returns the pairs this rank received."""
        import ctypes as C
        got = C.c_uint64()
        if self._lib.kct_rccl_merge_across_ranks(self._h, table._h, C.byref(got)) != 0:
            raise RuntimeError("kct_rccl_merge_across_ranks: " + self.last_error())
        return got.value

    def stats(self):
        import ctypes as C
        a, b, w = C.c_uint64(), C.c_uint64(), C.c_double()
        self._lib.kct_rccl_stats(self._h, C.byref(a), C.byref(b), C.byref(w))
        return {"bytes_sent": a.value, "bytes_received": b.value, "wait_seconds": w.value}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kct_rccl_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def merge_across_ranks(table, group=None, native=None):
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

    A ``store_kmers`` table's hash -> k-mer map (host side, as in the reference) follows its keys: the ranks' maps are gathered and
    every rank keeps the entries of the keys it now owns -- ``add()`` merges ``hash_to_kmer`` the same way (lib.rs:810-828).
    """
    import ctypes as C

    import numpy as np

    if native is not None and not getattr(table, "store_kmers", False):   # the whole exchange inside libkct_rccl.so (``NativeRccl``)
        return native.merge_across_ranks(table)
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    kmer_maps = None
    if getattr(table, "store_kmers", False):   # (a host-side dict per rank: small tables only, like everything store_kmers does)
        kmer_maps = [None] * world
        dist.all_gather_object(kmer_maps, dict(table._hash_to_kmer or {}), group=group)
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
    if kmer_maps is not None:
        merged = {}
        for m in kmer_maps:
            merged.update(m)
        table._hash_to_kmer = {int(h): merged[int(h)] for h in table.hashes if int(h) in merged}
    return recv.shape[0]


# ---- the EARLY route: super-k-mers travel to the GPU that owns them (csrc/kct_route.hip) ------------------------------------------
ROUTE_STATS = ("windows_sent", "windows_received", "bytes_sent", "bytes_received", "runs", "passes", "split_us", "exchange_wait_us",
               "owner_us", "region_retries", "window_starts")


class _Exchanger:
    """``kct_exchange_ops`` over ``torch.distributed``: device buffers come from torch (so that the collective can take them as
    tensors), the size exchange is an ``all_to_all_single`` of int64, the payload an ``all_to_all_single`` on bytes with uneven splits
    (RCCL over xGMI with the nccl backend, asynchronous: the library cuts and counts other passes meanwhile; staged through host
    memory, synchronously, when ranks share a GPU under gloo).  The library packs every peer's part back to back in rank order, which
    is the layout ``all_to_all_single`` wants."""

    def __init__(self, group, dev):
        import ctypes as C

        from ._lib import ExchangeOps
        self.group, self.dev = group, dev
        self.world = dist.get_world_size(group)
        self.host = dist.get_backend(group) != "nccl"
        self.keep = {}       # device address -> tensor
        self._size_bufs = {}
        self.host_s = 0.0    # seconds inside the callbacks OUTSIDE the collectives and staging copies (marshalling, bookkeeping)
        self.coll_s = 0.0    # seconds inside the collectives themselves (and, host-staged, their copies)
        self.calls = 0
        self.work = None
        self.pending = None  # (received host tensor, destination) of a host-staged exchange
        self.error = None
        self._cb = (ExchangeOps.ALLOC(self.alloc), ExchangeOps.RELEASE(self.release), ExchangeOps.SIZES(self.sizes), ExchangeOps.START(self.start),
                    ExchangeOps.WAIT(self.wait))
        self.ops = ExchangeOps(None, *self._cb)
        self.ptr = C.cast(C.pointer(self.ops), C.c_void_p)

    def _guard(self, fn, fail):
        try:
            return fn()
        except Exception as e:  # noqa: BLE001 -- reported through the library's status; raised by consume_device_early
            self.error = self.error or e
            return fail

    def alloc(self, _user, nbytes):
        def go():
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.dev)
            self.keep[t.data_ptr()] = t
            return t.data_ptr()
        return self._guard(go, None)

    def release(self, _user, p):
        self.keep.pop(int(p or 0), None)

    def sizes(self, _user, send, nvals, recv):
        def go():
            import time

            import numpy as np
            t0 = time.perf_counter()
            n = self.world * int(nvals)
            # (uint64 values travel as int64 bit patterns; the library's arrays are viewed in place, no per-element marshalling)
            src_np = np.ctypeslib.as_array(send, shape=(n,)).view(np.int64)
            key = (n, self.host)
            if key not in self._size_bufs:   # staging tensors kept across passes
                dev = "cpu" if self.host else self.dev
                self._size_bufs[key] = (torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.int64, device=dev),
                                        torch.empty(n, dtype=torch.int64, pin_memory=True) if not self.host else None)
            src, out, pinned = self._size_bufs[key]
            if self.host:
                src.copy_(torch.from_numpy(src_np))
                t1 = time.perf_counter()
                dist.all_to_all_single(out, src, group=self.group)
                t2 = time.perf_counter()
                vals = out.numpy()
            else:
                pinned.copy_(torch.from_numpy(src_np))
                t1 = time.perf_counter()
                src.copy_(pinned, non_blocking=True)
                dist.all_to_all_single(out, src, group=self.group)
                pinned.copy_(out)          # (synchronises with the collective on the current stream)
                t2 = time.perf_counter()
                vals = pinned.numpy()
            np.ctypeslib.as_array(recv, shape=(n,)).view(np.int64)[:] = vals
            self.coll_s += t2 - t1
            self.host_s += (t1 - t0) + (time.perf_counter() - t2)
            self.calls += 1
            return 0
        return self._guard(go, 1)

    def _view(self, base, nbytes):
        for addr, t in self.keep.items():
            if addr <= base and base + nbytes <= addr + t.numel():
                return t[base - addr: base - addr + nbytes]
        raise KeyError("the exchange was handed a buffer that did not come from its allocator")

    def start(self, _user, d_send, send_off, send_bytes, d_recv, recv_off, recv_bytes):
        def go():
            import time

            import numpy as np
            t0 = time.perf_counter()
            w = self.world
            arr = lambda p_: np.ctypeslib.as_array(p_, shape=(w,))  # noqa: E731
            sb, rb = arr(send_bytes).tolist(), arr(recv_bytes).tolist()
            assert arr(send_off).tolist() == np.concatenate([[0], np.cumsum(sb)[:-1]]).tolist() and arr(recv_off).tolist() == np.concatenate([[0], np.cumsum(rb)[:-1]]).tolist()
            src = self._view(int(d_send or 0), sum(sb)) if sum(sb) else torch.empty(0, dtype=torch.uint8, device=self.dev)
            dst = self._view(int(d_recv or 0), sum(rb)) if sum(rb) else torch.empty(0, dtype=torch.uint8, device=self.dev)
            t1 = time.perf_counter()
            if self.host:
                got = torch.empty(sum(rb), dtype=torch.uint8)
                dist.all_to_all_single(got, src.cpu(), output_split_sizes=rb, input_split_sizes=sb, group=self.group)
                dst.copy_(got)
                torch.cuda.synchronize()
            else:
                self.work = dist.all_to_all_single(dst, src, output_split_sizes=rb, input_split_sizes=sb, group=self.group, async_op=True)
            t2 = time.perf_counter()
            self.coll_s += t2 - t1
            self.host_s += t1 - t0
            return 0
        return self._guard(go, 1)

    def wait(self, _user):
        def go():
            if self.work is not None:
                self.work.wait()
                self.work = None
                torch.cuda.current_stream().synchronize()   # (the library counts on a stream of its own)
            return 0
        return self._guard(go, 1)


def _as_i64(values):
    import numpy as np
    return np.array(values, dtype=np.uint64).view(np.int64)


def consume_device_early(table, data_ptr, nbytes, consumed_bytes, group=None, max_windows=0, exchange_when_alone=False, native=None):
    """Counts this rank's device-resident record stream by the EARLY route (``kct_consume_device_routed``): every k-mer is counted by
    the rank that owns it -- owner = hash(minimiser) -- and what travels is super-k-mers: runs of consecutive windows with one owner as
    2-bit bases + a start bit per window.  Every rank must call it.  The owner side is the table's ordinary bulk path (``set_path``
    applies); tables need not agree on anything.  ``max_windows``: window starts per pass (0 = chosen by the library from free HBM;
    passes are pipelined: one is on the wire while the previous one is counted).  Returns (k-mers this rank counted as an owner,
    stats dict).

    Afterwards the ranks' tables are a disjoint partition of the key space: ``global_scalar_sum`` of ``len`` / ``sum_counts`` gives the
    global table's, and no ``merge_across_ranks`` is needed.  A failure on any rank raises on every rank.
    ``exchange_when_alone``: a group of ONE rank goes through the collectives too (everything is sent to itself) -- how the test suite
    runs the RCCL branch of the exchange on a one-GPU box.  ``native``: a ``NativeRccl`` -- the exchange is then libkct_rccl.so's
    (ncclSend / ncclRecv groups on its own stream, sizes marshalled in C), not ``torch.distributed``'s."""
    import ctypes as C

    if getattr(table, "store_kmers", False):
        raise ValueError("the early route moves packed bases only: a store_kmers table would lose its hash -> k-mer map")
    n, stats = C.c_uint64(), (C.c_uint64 * 16)()
    if native is not None:
        st = table._lib.kct_consume_device_routed(table._h, C.c_void_p(int(data_ptr)), int(nbytes), int(consumed_bytes), native.world, native.rank,
                                                  native.ops if native.world > 1 or exchange_when_alone else None, int(max_windows), C.byref(n), stats)
        table._check(st)
        return n.value, dict(zip(ROUTE_STATS, (int(v) for v in stats)))
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    ex = _Exchanger(group, dev) if world > 1 or exchange_when_alone else None
    consume_device_early.last_exchanger = ex   # (tests look at what it still holds after a failed call)
    st = table._lib.kct_consume_device_routed(table._h, C.c_void_p(int(data_ptr)), int(nbytes), int(consumed_bytes), world, rank,
                                              ex.ptr if ex else None, int(max_windows), C.byref(n), stats)
    if ex is not None and ex.error is not None:
        raise ex.error
    table._check(st)
    out = dict(zip(ROUTE_STATS, (int(v) for v in stats)))
    if ex is not None:   # the Python glue's own cost: per size exchange, outside the collectives
        out["glue_host_us_per_exchange"] = round(ex.host_s * 1e6 / max(1, ex.calls), 1)
        out["glue_collective_us"] = round(ex.coll_s * 1e6, 1)
        out["glue_size_exchanges"] = ex.calls
    return n.value, out
