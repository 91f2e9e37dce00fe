"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports exactly
what include/*.h declares; the Python shim fails loudly without a GPU (no CPU fallback)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for hdr in ("kct.h", "kct_synth.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(kct_[a-z_0-9]+)\s*\(", text))
    return names


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oxli_amd", "csrc")], check=True)
    from oxli_amd import _lib
    return _lib


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "kct.h"\n#include "kct_synth.h"\nint main(void){ kct_table *t = 0; (void)t; return KCT_OK; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                    str(tmp_path / "t.o")], check=True)


def test_library_exports_every_declared_symbol(lib):
    handle = lib.load()
    declared = _declared()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(handle, name), f"libkct_hip.so lacks {name}"
    # and the binding table covers the header exactly (no stale or missing prototypes)
    assert set(lib.SIGNATURES) == declared


def test_library_exports_nothing_else(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    exported -= {"_init", "_fini"}
    assert exported == _declared()   # built with -fvisibility=hidden: the C ABI is the only code the library exports


def test_rccl_exchange_library_exports_its_header(lib, tmp_path):
    """include/kct_rccl.h: kct_exchange_ops over RCCL, a library of its own (libkct_hip.so must not depend on RCCL)."""
    import ctypes
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "kct_rccl.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(kct_rccl_[a-z_0-9]+)\s*\(", text))
    assert {"kct_rccl_unique_id", "kct_rccl_create", "kct_rccl_ops", "kct_rccl_destroy", "kct_rccl_merge_across_ranks"} <= declared
    path = os.path.join(ROOT, "oxli_amd", "csrc", "libkct_rccl.so")
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l} - {"_init", "_fini"}
    assert exported == declared
    handle = ctypes.CDLL(path)
    assert all(hasattr(handle, n) for n in declared)
    from oxli_amd import _lib as L_
    assert set(L_.RCCL_SIGNATURES) == declared      # the ctypes table mirrors the header
    needed = subprocess.run(["readelf", "-d", lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()
    src = tmp_path / "t.c"
    src.write_text('#include "kct_rccl.h"\nint main(void){ kct_rccl *x = 0; (void)x; return KCT_RCCL_ID_BYTES == 128 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)


def test_library_contains_gfx950_code_only(lib):
    data = open(lib.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", data))
    assert targets == {b"gfx950"}, targets


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    handle = lib.load()
    assert handle.kct_device_count() == 0
    import oxli_amd
    with pytest.raises(RuntimeError, match="no HIP device"):
        oxli_amd.KmerCountTable(21)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "oxli_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "kct_oracle" not in text, f


def test_call_glue_is_built_and_bound():
    """csrc/pyfast.c: the CPython shim per-record consume() loops go through (it only forwards to kct_consume)."""
    import oxli_amd.table as T
    assert T._fast_consume is not None, "oxli_amd/_kctfast.so missing: run `make -C oxli_amd/csrc`"
    assert T._fast.consume(0, 12345, True) is None  # neither str nor bytes: left to the ctypes path (no call is made)
    with pytest.raises(TypeError):
        T._fast.consume(0, "ACGT")
    import numpy as np
    seqs = ["ACGT" * 37 + "AC", b"GGGTTT", "", "\u00e9", "N" * 5]
    data, offs = T._fast.csr(seqs)  # what consume_batch(list) hands to kct_consume_batch
    parts = [T._bytes(x) for x in seqs]
    assert data == b"".join(parts)
    assert np.frombuffer(offs, dtype=np.uint64).tolist() == [0] + list(np.cumsum([len(x) for x in parts]))
    assert T._fast.csr(tuple(seqs)) == (data, offs) and T._fast.csr([]) == (b"", bytes(8))
    assert T._fast.csr(["A", bytearray(b"C")]) is None  # left to the Python path
    text = open(os.path.join(ROOT, "oxli_amd", "csrc", "pyfast.c")).read()
    assert "hip" not in text.replace("libkct_hip", "") and "murmur" not in text.lower()  # glue only
