"""The hash-0 corner of consume (lib.rs:589: `Ok(0) => continue` -- such a window is neither counted nor tallied), run instead of
argued: `make zero` builds the library with ONE chosen 21-mer hashing to 0 on the device (kmer_device.h hash_packed), and the worker
drives every counting path of that build against the oracle's table minus that k-mer."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_kmer_that_hashes_to_zero_is_skipped_on_every_path():
    csrc = os.path.join(ROOT, "oxli_amd", "csrc")
    so = os.path.join(csrc, "libkct_zero.so")
    hdrs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip"))]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in hdrs):
        subprocess.run(["make", "-j8", "-C", csrc, "zero"], check=True, capture_output=True, timeout=1500)   # (build() does this ahead of time)
    env = dict(os.environ, KCT_LIB_PATH=so)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "zero_hash_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + "\n" + out.stderr[-6000:]
    assert "ZERO_HASH_OK" in out.stdout
