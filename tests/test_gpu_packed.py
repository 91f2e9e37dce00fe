"""Packed base arrays (include/kct.h: 2-bit codes + validity bits, sixteen bases per group) as the input format: the device
packer against a plain numpy encoding, ``kct_consume_device_packed`` against the ASCII stream and the oracle on every path,
and ``kct_consume_batch``'s packed upload (host SIMD packer, 0.375 B per base over PCIe) against its ASCII upload and the
oracle -- lower case, N, non-ASCII bytes, records shorter than k, empty records (lib.rs:548 byte semantics, 576-600)."""
import ctypes as C
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import OracleTable  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    import torch
    from oxli_amd import KmerCountTable, _lib
    return torch, KmerCountTable, _lib.load()


def messy_records(rng, n, lo=0, hi=400):
    alphabet = "ACGT" * 30 + "acgt" * 4 + "N" + "n" + "R" + "é" + "-"
    recs = []
    for _ in range(n):
        L = rng.randint(lo, hi)
        recs.append("".join(rng.choice(alphabet) for _ in range(L)) if rng.random() < 0.3 else "".join(rng.choice("ACGT") for _ in range(L)))
    return recs


def np_pack(stream):
    """Reference encoding of a byte stream: codes (first base in bits 31:30) and validity (first base in bit 15) per 16 bytes."""
    b = np.frombuffer(stream + b"\n" * ((-len(stream)) % 16), dtype=np.uint8).reshape(-1, 16)
    low = b | 0x20
    ok = (low == ord("a")) | (low == ord("c")) | (low == ord("g")) | (low == ord("t"))
    x = (b >> 1) & 3
    x = x ^ (x >> 1)
    x = np.where(ok, x, 0).astype(np.uint32)
    shifts = np.arange(30, -2, -2, dtype=np.uint32)
    codes = (x << shifts).sum(axis=1).astype(np.uint32)
    valid = (ok.astype(np.uint32) << np.arange(15, -1, -1, dtype=np.uint32)).sum(axis=1).astype(np.uint16)
    return codes, valid


def test_device_packer_matches_numpy(gpu):
    torch, _, lib = gpu
    rng = random.Random(3)
    stream = ("\n".join(messy_records(rng, 3000)) + "\n").encode("utf-8")
    dev = torch.frombuffer(bytearray(stream + b"\n" * ((-len(stream)) % 16)), dtype=torch.uint8).cuda()
    ng = (len(stream) + 15) // 16
    codes = torch.zeros(ng, dtype=torch.int32, device="cuda")
    valid = torch.zeros(ng, dtype=torch.int16, device="cuda")
    assert lib.kct_pack_stream_device(dev.data_ptr(), len(stream), codes.data_ptr(), valid.data_ptr(), None) == 0
    torch.cuda.synchronize()
    rc, rv = np_pack(stream)
    got_v = valid.cpu().numpy().view(np.uint16)
    assert np.array_equal(got_v, rv)
    # (codes under invalid bases are unspecified on the device: compare the valid ones)
    mask = np.repeat(((rv[:, None] >> np.arange(15, -1, -1, dtype=np.uint16)) & 1).astype(np.uint32), 1, axis=1)
    keep = (mask * np.uint32(3) << np.arange(30, -2, -2, dtype=np.uint32)).sum(axis=1).astype(np.uint32)
    assert np.array_equal(codes.cpu().numpy().view(np.uint32) & keep, rc & keep)


@pytest.mark.parametrize("k,path", [(21, "auto"), (21, "dedupe"), (21, "partitioned"), (21, "direct"), (31, "dedupe"), (31, "partitioned"),
                                    (51, "partitioned"), (64, "auto"), (1, "auto"), (5, "direct"), (90, "partitioned")])
def test_packed_device_stream_equals_ascii_and_oracle(gpu, k, path):
    torch, KCT, lib = gpu
    rng = random.Random(100 + k)
    recs = messy_records(rng, 20000, 0, 300) + ["".join(rng.choice("ACGT") for _ in range(600_000))] + messy_records(rng, 20000, 100, 200)
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    stream = ("\n".join(recs) + "\n").encode("utf-8")
    dev = torch.frombuffer(bytearray(stream + b"\n" * ((-len(stream)) % 16)), dtype=torch.uint8).cuda()
    ng = (len(stream) + 15) // 16
    codes = torch.zeros(ng, dtype=torch.int32, device="cuda")
    valid = torch.zeros(ng, dtype=torch.int16, device="cuda")
    assert lib.kct_pack_stream_device(dev.data_ptr(), len(stream), codes.data_ptr(), valid.data_ptr(), None) == 0
    torch.cuda.synchronize()
    consumed = sum(len(r.encode("utf-8")) for r in recs)
    rk, rc = ref.dump_arrays()
    for packed in (True, False):
        t = KCT(k, capacity=3_000_000)
        t.set_path(path)
        n = t.consume_device_packed(codes.data_ptr(), valid.data_ptr(), len(stream), consumed) if packed else \
            t.consume_device(dev.data_ptr(), len(stream), consumed)
        assert n == n_ref, (packed, n, n_ref)
        dk, dc = t.dump_arrays(1)
        assert np.array_equal(dk, rk) and np.array_equal(dc, rc), packed
        assert t.consumed == ref.consumed


@pytest.mark.parametrize("k,path", [(21, "dedupe"), (21, "partitioned"), (31, "dedupe"), (51, "partitioned"), (25, "partitioned"), (25, "dedupe"), (15, "dedupe")])
def test_packed_arrays_cut_mid_group_and_misaligned(gpu, k, path):
    """K1's PACKED instantiations fetch a tile's words sixteen bytes per lane (k1_kernel.h): the stream's last group is cut to nbases
    inside the loading lane, pieces that would reach past the arrays are fetched word by word, and arrays that are not 16-byte
    aligned (here: the same arrays entered one group further in) go through the ordinary instantiation's narrow loads.  All of it
    against the ASCII stream cut at the same byte and the oracle.  (k = 25: no PACKED instantiation of its own -- run-time k, the hashing
    mode and the 64-bit dedupe-first mode k1_packed<1, 0, 1>; k = 15: the compact dedupe-first mode at run-time k, k1_packed<1, 0, 2>.)"""
    torch, KCT, lib = gpu
    rng = random.Random(500 + k)
    body = "".join(rng.choice("ACGT") for _ in range(3_000_000))          # one long record: every cut lands among valid bases
    stream = (body + "\n").encode()
    dev = torch.frombuffer(bytearray(stream + b"\n" * ((-len(stream)) % 16) + b"\n" * 64), dtype=torch.uint8).cuda()
    ng = (len(stream) + 15) // 16
    codes = torch.zeros(ng + 4, dtype=torch.int32, device="cuda")
    valid = torch.zeros(ng + 8, dtype=torch.int16, device="cuda")
    assert lib.kct_pack_stream_device(dev.data_ptr(), len(stream), codes.data_ptr(), valid.data_ptr(), None) == 0
    torch.cuda.synchronize()
    for cut, skip in ((len(stream) - 7, 0), (len(stream) - 16 * 1000 - 3, 0), (16384 * 3 + 5, 0), (len(stream) - 9, 1), (16384 * 2 + 16 * 5 + 11, 3)):
        # `skip` groups dropped in front: the arrays then start 4 * skip / 2 * skip bytes into their allocation (misaligned for 16-byte loads)
        nb = cut - 16 * skip
        text = body[16 * skip:cut]
        ref = OracleTable(k)
        n_ref = ref.consume(text)
        rk, rc = ref.dump_arrays()
        t = KCT(k, capacity=6_000_000)
        t.set_path(path)
        n = t.consume_device_packed(codes.data_ptr() + 4 * skip, valid.data_ptr() + 2 * skip, nb, nb)
        assert n == n_ref, (cut, skip, n, n_ref)
        dk, dc = t.dump_arrays(1)
        assert np.array_equal(dk, rk) and np.array_equal(dc, rc), (cut, skip)
        a = KCT(k, capacity=6_000_000)
        a.set_path(path)
        assert a.consume_device(dev.data_ptr() + 16 * skip, nb, nb) == n_ref
        ak, ac = a.dump_arrays(1)
        assert np.array_equal(ak, rk) and np.array_equal(ac, rc)


@pytest.mark.parametrize("k", [21, 31, 51])
def test_batch_packed_upload_equals_ascii_upload_and_oracle(gpu, k):
    """kct_consume_batch with >= 8 MiB of records: the host packers (16 pool threads, SSSE3) + packed H2D, against the ASCII
    upload and the oracle.  Ragged records, empty ones, lower case, N, multi-byte UTF-8."""
    torch, KCT, _ = gpu
    rng = random.Random(7 + k)
    recs = messy_records(rng, 60000, 0, 400) + [""] * 50 + ["".join(rng.choice("ACGT") for _ in range(150)) for _ in range(30000)] + \
        ["".join(rng.choice("ACGTacgtN") for _ in range(2_000_003))]
    rng.shuffle(recs)
    assert sum(len(r) for r in recs) > (9 << 20)
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    rk, rc = ref.dump_arrays()
    for packed in (True, False):
        t = KCT(k, capacity=8_000_000)
        t.set_packed_upload(packed)
        t.profile(True)
        assert t.consume_batch(recs) == n_ref
        tl = t.batch_timeline()     # kct_batch_timeline: filled by the packed-upload route only
        if packed:
            assert tl["threads"] >= 2 and tl["source_bytes"] == sum(len(r.encode()) for r in recs) and tl["packed_bytes"] > 0
            assert 0 <= tl["first_packer_start_ms"] <= tl["last_packer_end_ms"] <= tl["submitted_ms"] and tl["threads_busy_ms_sum"] >= tl["thread_busy_ms_max"] > 0
        else:
            assert tl["threads"] == 0 and tl["submitted_ms"] == 0
        dk, dc = t.dump_arrays(1)
        assert np.array_equal(dk, rk) and np.array_equal(dc, rc), packed
        assert t.consumed == ref.consumed
        # error mode never packs (the first bad byte is found on the ASCII stream): same answer either way
        with pytest.raises(ValueError) as e1:
            t.consume_batch(recs, skip_bad_kmers=False)
        r2 = OracleTable(k)
        with pytest.raises(ValueError) as e2:
            for r in recs:
                r2.consume(r, skip_bad_kmers=False)
        assert str(e1.value) == str(e2.value)


@pytest.mark.parametrize("k,flushers,cap", [(21, 4, 400_000), (21, 2, 400_000), (17, 4, 400_000), (21, 4, 40_000_000)])
def test_wave_specialised_k1_equals_oracle(gpu, k, flushers, cap, monkeypatch):
    """KCT_K1_FLUSHERS (k1ws_kernel.h): the compact dedupe-first K1 with hashing waves that never reach a workgroup barrier and flusher
    waves that move complete lines out after a grace period -- not the default, but it must count exactly what partition_windows_kernel
    counts: ASCII and packed input, messy records (holes, the overflow route: a homopolymer run hammers one bin), a one-level and a
    two-level shadow (capacity 4x10^7: 2^26 slots), against the oracle."""
    torch, KCT, lib = gpu
    rng = random.Random(900 + k + flushers)
    genome = "".join(rng.choice("ACGT") for _ in range(200_000))
    recs = [genome[i:i + 150] for i in (rng.randrange(0, len(genome) - 150) for _ in range(40_000))] + messy_records(rng, 4000, 0, 300) + ["A" * 5000, "ACGT" * 2000, ""]
    rng.shuffle(recs)
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    rk, rc = ref.dump_arrays()
    stream = "".join(r + "\n" for r in recs).encode()
    dev = torch.frombuffer(bytearray(stream + b"\n" * ((-len(stream)) % 16) + b"\n" * 64), dtype=torch.uint8).cuda()
    ng = (len(stream) + 15) // 16
    codes = torch.zeros(ng + 4, dtype=torch.int32, device="cuda")
    valid = torch.zeros(ng + 8, dtype=torch.int16, device="cuda")
    assert lib.kct_pack_stream_device(dev.data_ptr(), len(stream), codes.data_ptr(), valid.data_ptr(), None) == 0
    torch.cuda.synchronize()
    monkeypatch.setenv("KCT_K1_FLUSHERS", str(flushers))   # (read by kct_create)
    for packed in (False, True):
        t = KCT(k, capacity=cap)
        t.set_path("dedupe")
        t.profile(True)
        consumed = sum(len(r) for r in recs)
        n = t.consume_device_packed(codes.data_ptr(), valid.data_ptr(), len(stream), consumed) if packed else t.consume_device(dev.data_ptr(), len(stream), consumed)
        assert n == n_ref
        n += t.consume_device(dev.data_ptr(), len(stream), consumed)     # a second pass meets a live shadow
        dk, dc = t.dump_arrays(1)
        prof = t.profile_read()
        assert "partition_windows_kernel<compact>" in prof, prof
        assert np.array_equal(dk, rk) and np.array_equal(dc, 2 * rc), (packed,)
        assert t.sum_counts == 2 * n_ref


@pytest.mark.parametrize("k,path", [(21, "dedupe"), (17, "dedupe"), (31, "dedupe"), (31, "partitioned"), (25, "partitioned"), (51, "partitioned")])
def test_position_sorted_reads_switch_k1_to_the_short_flush_interval(gpu, k, path):
    """Position-sorted reads bring every k-mer ~30 times within a few hundred windows: the compact K1's rings overflow between two flushes (6.6 % of
    C2 sorted), the pass notices (more than 2 % over the overflow route) and the table's later passes run the FE = 4 instantiation of K1
    (k1_kernel.h: a flush every 4 windows; k = 21 at compile time, k = 17 at run time; ASCII and packed input) -- and the 8-byte-entry modes likewise
    (a flush every 2 windows: the 64-bit dedupe-first mode at k = 31, the hashing mode at k = 31 / 51 at compile time and k = 25 at run time).
    Every pass must equal the oracle."""
    torch, KCT, lib = gpu
    rng = random.Random(4000 + k)
    genome = "".join(rng.choice("ACGT") for _ in range(330_000))
    recs = [genome[p:p + 150] for p in range(0, len(genome) - 150, 5)]          # 66 k reads, sorted, 30x
    ref = OracleTable(k)
    n_ref = sum(ref.consume(r) for r in recs)
    rk, rc = ref.dump_arrays()
    stream = "".join(r + "\n" for r in recs).encode()
    dev = torch.frombuffer(bytearray(stream + b"\n" * ((-len(stream)) % 16) + b"\n" * 64), dtype=torch.uint8).cuda()
    ng = (len(stream) + 15) // 16
    codes = torch.zeros(ng + 4, dtype=torch.int32, device="cuda")
    valid = torch.zeros(ng + 8, dtype=torch.int16, device="cuda")
    assert lib.kct_pack_stream_device(dev.data_ptr(), len(stream), codes.data_ptr(), valid.data_ptr(), None) == 0
    torch.cuda.synchronize()
    consumed = sum(len(r) for r in recs)
    t = KCT(k, capacity=5_000_000)   # (2^23 slots = 1024 blocks: K1 fans out to 1024 bins of 32 / 16 ring entries, as for C2)
    t.set_path(path)
    t.profile(True)
    total = 0
    for i in range(4):   # pass 0 overflows and sets the switch; passes 1-3 run with the short interval (ASCII, packed, ASCII)
        n = t.consume_device_packed(codes.data_ptr(), valid.data_ptr(), len(stream), consumed) if i == 2 else t.consume_device(dev.data_ptr(), len(stream), consumed)
        assert n == n_ref
        total += n
        dk, dc = t.dump_arrays(1)
        assert np.array_equal(dk, rk) and np.array_equal(dc, (i + 1) * rc), i
    prof = t.profile_read()
    assert "merge_overflow_kernel" in prof and t.sum_counts == total
