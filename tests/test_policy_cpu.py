"""Path selection and sizing rules of the counting paths (oxli_amd/csrc/path_policy.h: pure host logic, shared with
libkct_hip.so) driven with fake table geometries on the CPU -- no GPU, no HIP: tests/policy_harness.cpp instantiates the
header's templates for a plain struct and is built here with g++."""
import ctypes as C
import math
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIRECT, PARTITIONED, DEDUPE64, COMPACT = 0, 1, 2, 3


class PolicyIn(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("k", "block_bits", "force_path", "dedupe_off", "compact_off", "dedupe_hint", "auto_sized", "shadow_dirty", "s32_dirty")] + \
               [(n, C.c_uint64) for n in ("cap", "n_keys", "shadow_keys", "s32_keys", "windows_since_read", "call_windows_left")]


@pytest.fixture(scope="module")
def pol(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("policy") / "libpolicy.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Werror", "-o", so, os.path.join(ROOT, "tests", "policy_harness.cpp")], check=True)
    lib = C.CDLL(so)
    lib.policy_choose_path.argtypes = [C.POINTER(PolicyIn), C.c_uint64]
    lib.policy_probe_wanted.argtypes = [C.POINTER(PolicyIn), C.c_uint64]
    lib.policy_mostly_new_expected.argtypes = [C.POINTER(PolicyIn), C.c_uint64]
    lib.policy_probe_verdict.argtypes = [C.POINTER(PolicyIn), C.c_double, C.c_uint64]
    lib.policy_draws_per_distinct.restype = C.c_double
    lib.policy_draws_per_distinct.argtypes = [C.c_double]
    lib.policy_per_distinct_two_depths.restype = C.c_double
    lib.policy_per_distinct_two_depths.argtypes = [C.c_double] * 5
    lib.policy_region_capacity.restype = C.c_uint
    lib.policy_region_capacity.argtypes = [C.c_double]
    lib.policy_overflow_capacity.restype = C.c_uint
    lib.policy_overflow_capacity.argtypes = [C.c_uint64]
    lib.policy_min_lines.restype = C.c_uint
    return lib


def table(k=21, cap=1 << 23, **kw):
    t = PolicyIn(k=k, cap=cap, block_bits=min(13, int(math.log2(cap))))
    for n, v in kw.items():
        setattr(t, n, v)
    return t


C2_WINDOWS = 1_000_000 * 151


def test_headline_shapes(pol):
    """C2 (2^23 slots = 1024 blocks, one level)."""
    choose = lambda t, n: pol.policy_choose_path(C.byref(t), n)  # noqa: E731
    cold = table()
    assert choose(cold, C2_WINDOWS) == PARTITIONED                       # knows nothing, no hint: hash (unless the probe says otherwise)
    assert pol.policy_probe_wanted(C.byref(cold), C2_WINDOWS) == 1       # ... and a call of 1.5x10^8 windows is probed first
    assert choose(table(dedupe_hint=1), C2_WINDOWS) == COMPACT           # the steady state: cleared table, hint kept
    assert choose(table(k=31, dedupe_hint=1), C2_WINDOWS) == DEDUPE64
    assert choose(table(k=51), C2_WINDOWS) == PARTITIONED                 # k > 32, nothing known: hash every window
    assert choose(table(k=51, dedupe_hint=1), C2_WINDOWS) == PARTITIONED  # the 128-bit dedupe-first variant is not chosen by itself (round 4:
    assert choose(table(k=51, n_keys=2_000_000), C2_WINDOWS) == PARTITIONED   # 1.03x on its showcase) ... only when forced, below
    assert choose(table(k=51, n_keys=5_000_000), C2_WINDOWS) == PARTITIONED   # more k-mers than 1024 x 4096 slots take
    assert choose(table(k=51, n_keys=2_000_000), 40_000_000) == PARTITIONED   # 20 per known k-mer: not worth a conversion by atomics
    assert choose(table(k=65, dedupe_hint=1), C2_WINDOWS) == PARTITIONED
    assert choose(table(k=47, n_keys=2_000_000), C2_WINDOWS) == PARTITIONED and choose(table(k=64, n_keys=2_000_000), C2_WINDOWS) == PARTITIONED
    assert choose(table(n_keys=5_000_000), C2_WINDOWS) == COMPACT        # 30 windows per known k-mer ahead
    assert choose(table(n_keys=5_000_000), 40_000_000) == PARTITIONED    # 8 per known k-mer: not worth a conversion
    assert choose(table(n_keys=5_000_000, windows_since_read=10 ** 9), 40_000_000) == COMPACT   # ... unless reads are rare
    assert choose(table(n_keys=5_000_000, dedupe_off=1), C2_WINDOWS) == PARTITIONED
    assert choose(table(n_keys=5_000_000, compact_off=1), C2_WINDOWS) == DEDUPE64
    assert choose(table(), 500_000) == DIRECT and choose(table(cap=1 << 16), C2_WINDOWS) == DIRECT   # small pass / tiny table
    for force, want in ((1, DIRECT), (2, PARTITIONED), (3, COMPACT)):
        assert choose(table(force_path=force), C2_WINDOWS) == want
    assert choose(table(force_path=3, k=31), C2_WINDOWS) == DEDUPE64 and choose(table(force_path=3, k=41), C2_WINDOWS) == 4
    assert choose(table(force_path=3, k=90), C2_WINDOWS) == PARTITIONED


def test_a_shadow_must_be_paid_for(pol):
    """Two-level tables: the compact shadow is half the table's bytes (as many blocks as the table: since round 4 also below 2^16
    blocks), the 64-bit one table-sized -- 0.15 windows per shadow byte."""
    choose = lambda t, n: pol.policy_choose_path(C.byref(t), n)  # noqa: E731
    sub1 = table(cap=1 << 26, dedupe_hint=1)                             # C2 with 1 % substitution errors: 2^26 slots
    assert pol.policy_compact_sbits(C.byref(sub1)) == 13                 # 8192 blocks, 512 MiB (it was rounded up to 2^16 blocks = 4 GiB)
    assert choose(sub1, C2_WINDOWS) == COMPACT                           # 1.5x10^8 windows pay for 512 MiB of shadow ...
    assert choose(sub1, 60_000_000) == PARTITIONED                       # ... 6x10^7 do not (nor for 1 GiB of 64-bit shadow)
    assert pol.policy_probe_wanted(C.byref(table(cap=1 << 26)), C2_WINDOWS) == 0    # 2.2 window starts per slot: mostly distinct k-mers, no need to look
    assert pol.policy_probe_wanted(C.byref(table(cap=1 << 25)), C2_WINDOWS) == 1    # 4.5 per slot: look
    # ... and a call of fewer than 2.5 per slot into an empty table its owner sized is taken for mostly first sightings (K2 claims slots)
    assert pol.policy_mostly_new_expected(C.byref(table(cap=1 << 26)), C2_WINDOWS) == 1
    assert pol.policy_mostly_new_expected(C.byref(table(cap=1 << 30)), 1_900_000_000) == 1       # C4's shard
    assert pol.policy_mostly_new_expected(C.byref(table(cap=1 << 25)), C2_WINDOWS) == 0
    assert pol.policy_mostly_new_expected(C.byref(table(cap=1 << 26, n_keys=1)), C2_WINDOWS) == 0
    assert pol.policy_mostly_new_expected(C.byref(table(cap=1 << 26, auto_sized=1)), C2_WINDOWS) == 0
    assert pol.policy_compact_sbits(C.byref(table(cap=1 << 23))) == 10 and pol.policy_compact_sbits(C.byref(table(cap=1 << 24))) == 11
    assert pol.policy_compact_sbits(C.byref(table(cap=1 << 30))) == 17
    assert choose(table(cap=1 << 24, dedupe_hint=1), 90_000_000) == COMPACT    # 128 MiB of compact shadow: paid by 2x10^7 windows
    assert choose(table(cap=1 << 24, dedupe_hint=1, compact_off=1), 90_000_000) == DEDUPE64   # 256 MiB of 64-bit shadow: by 4x10^7
    assert choose(table(cap=1 << 26, dedupe_hint=1, s32_dirty=1), 60_000_000) == COMPACT   # counts already pending: the shadow is paid for
    ns = table(cap=1 << 30)                                              # the north-star run: 1.5x10^10 windows, 8 GiB of compact shadow
    assert pol.policy_probe_wanted(C.byref(ns), 15_100_000_000) == 1
    assert pol.policy_probe_verdict(C.byref(ns), 26.0, 15_100_000_000) == 1 and pol.policy_probe_verdict(C.byref(ns), 3.3, 15_100_000_000) == 0
    assert choose(table(cap=1 << 30, dedupe_hint=1), 15_100_000_000) == COMPACT
    assert choose(table(cap=1 << 30, k=31, dedupe_hint=1), 7_500_000_000) == DEDUPE64
    assert choose(table(cap=1 << 34, dedupe_hint=1), 1 << 34) == DIRECT   # beyond two levels of 1024 bins


def test_probe_preconditions(pol):
    want = lambda t, n: pol.policy_probe_wanted(C.byref(t), n)  # noqa: E731
    assert want(table(), 8 << 22) == 1 and want(table(), (8 << 22) - 1) == 0
    for off in (dict(dedupe_hint=1), dict(dedupe_off=1), dict(auto_sized=1), dict(n_keys=1), dict(s32_keys=1), dict(force_path=2), dict(k=33), dict(cap=1 << 22)):
        assert want(table(**off), C2_WINDOWS) == 0, off


def test_draws_per_distinct_inverts_the_sampling_law(pol):
    for x in (0.05, 0.5, 1.0, 3.3, 26.0, 300.0):
        r = (1.0 - math.exp(-x)) / x
        assert pol.policy_draws_per_distinct(r) == pytest.approx(x, rel=1e-6)
    assert pol.policy_draws_per_distinct(1.0) == 0.0 and pol.policy_draws_per_distinct(0.0) == 1e9


def test_sizing_rules(pol):
    out = (C.c_int * 6)()
    expect = {8: (8, 0, 0, 256, 1), 10: (10, 0, 0, 1024, 1), 11: (5, 6, 1, 32, 8), 14: (8, 6, 1, 256, 1), 15: (8, 7, 1, 256, 1), 17: (10, 7, 1, 1024, 1), 20: (10, 10, 1, 1024, 1)}
    for bbits, (pbits, sub, two, P, W) in expect.items():
        pol.policy_levels(bbits, 256, out)
        assert tuple(out[:5]) == (pbits, sub, two, P, W), (bbits, tuple(out))
        assert out[0] + out[1] == bbits
    for avg in (0.0, 10.0, 944.0, 1e5):
        cap = pol.policy_region_capacity(avg)
        assert cap % 8 == 0 and cap >= avg * 1.15 + 64
    assert pol.policy_overflow_capacity(1000) == 4096 and pol.policy_overflow_capacity(1 << 20) == 1 << 17 and pol.policy_overflow_capacity(1 << 30) == 1 << 20
    assert pol.policy_min_lines(16384, 6, 8) == 4 and pol.policy_min_lines(16384, 8, 8) == 2 and pol.policy_min_lines(16384, 10, 8) == 1
    assert pol.policy_min_lines(32768, 7, 4) == 4 and pol.policy_min_lines(8192, 7, 16) == 4


def test_two_depth_estimate_sees_never_repeating_k_mers(pol):
    """The dry probe's estimator (two depths of one sample): uniform draws from D k-mers plus a share e of k-mers that never repeat
    (sequencing errors).  On its own model it recovers N / F(N); with e = 0 it agrees with the one-depth law."""
    import math

    def F(n, e, D):
        return e * n + D * (1.0 - math.exp(-(1.0 - e) * n / D))
    n1, n2 = 2_000_000.0, 4_000_000.0
    for e, D, N in ((0.0, 5e6, 1.5e8), (0.2, 5e6, 1.5e8), (0.25, 5e8, 3.3e9), (0.0, 5e8, 1.5e10), (0.02, 4e6, 1.5e8), (0.5, 1e6, 1e8)):
        got = pol.policy_per_distinct_two_depths(n1, F(n1, e, D), n2, F(n2, e, D), N)
        assert got == pytest.approx(N / F(N, e, D), rel=0.03), (e, D, N, got)
    # C2 with 1 % substitutions (bench.py C2_sub1pct): the truth is 4.5 per distinct k-mer, not the one-depth estimate's 17
    assert pol.policy_per_distinct_two_depths(n1, F(n1, 0.19, 5e6), n2, F(n2, 0.19, 5e6), 1.3e8) < 6.0
    assert pol.policy_per_distinct_two_depths(n1, F(n1, 0.0, 5e6), n2, F(n2, 0.0, 5e6), 1.3e8) == pytest.approx(26.0, rel=0.02)
