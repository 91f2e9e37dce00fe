#!/usr/bin/env python3
"""Regenerates the derived golden fixtures in this directory.  Run from the repo root, in the
build container (it reads /root/reference and a third-party MurmurHash3 source on disk):

    python tests/golden/make_golden.py

What it writes
  murmur3_x64_128_vectors.json  h1 of MurmurHash3_x64_128(seed 42) for ACGT strings of length
        0..70 and a few named inputs, computed by an INDEPENDENT canonical implementation:
        Appleby's public-domain MurmurHash3.cpp as shipped inside scikit-learn
        (sklearn/utils/src/MurmurHash3.cpp), compiled here with g++.  This pins the 16-byte
        block loop (k >= 16), which none of the reference's own known answers (all k <= 5)
        exercises.  It is NOT oxli source and is not copied into the repo.
  example_fa_digests.json       n / distinct / max / histogram head / first hashes / XOR and
        SUM checksums / SHA-256 of the `dump(sortkeys=True)` TSV text for doc/example.fa at
        k = 21, 31, 51, computed by oracle/kct_oracle.c AFTER that oracle has passed the
        reference's known answers (reference_kats.json) and the vectors above.  Labelled
        "derived by the validated restatement", not by running the Rust reference.
  example.fa                    data file copied from the reference's doc/example.fa.

The reference is Rust and cannot be imported or built here (no cargo/rustc), so no fixture in
this directory was produced by executing reference code.
"""
import ctypes as C
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SKL = "/usr/local/lib/python3.10/dist-packages/sklearn/utils/src"


def canonical_murmur():
    tmp = tempfile.mkdtemp()
    so = os.path.join(tmp, "libmm3.so")
    shim = os.path.join(tmp, "shim.cpp")
    with open(shim, "w") as f:
        f.write('#include "MurmurHash3.h"\n#include <cstdint>\n'
                'extern "C" uint64_t mm3_h1(const void* p, int n, uint32_t seed)'
                '{ uint64_t o[2]; MurmurHash3_x64_128(p, n, seed, o); return o[0]; }\n')
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-I", SKL, shim, os.path.join(SKL, "MurmurHash3.cpp"), "-o", so],
                   check=True)
    lib = C.CDLL(so)
    lib.mm3_h1.restype = C.c_uint64
    lib.mm3_h1.argtypes = [C.c_char_p, C.c_int, C.c_uint32]
    return lambda b: lib.mm3_h1(b, len(b), 42)


def read_fasta(path):
    return "".join(l.strip() for l in open(path) if not l.startswith(">"))


def main():
    import oracle

    mm3 = canonical_murmur()
    rng = random.Random(20261002)
    vecs = []
    for n in range(0, 71):
        for _ in range(3):
            s = "".join(rng.choice("ACGT") for _ in range(n))
            vecs.append({"bytes": s, "h1": mm3(s.encode())})
    for s in ["ACGTACGTACGTACGT", "AAATCTTATAAAATAACCACA", "TAAACCCTAACCCTAACCCTAACCCTAACCC",
              "N" * 255, "ACGT" * 63 + "ACG"]:
        vecs.append({"bytes": s, "h1": mm3(s.encode())})
    bad = [v for v in vecs if oracle.murmur64(v["bytes"]) != v["h1"]]
    assert not bad, bad[:3]
    json.dump({"_comment": "MurmurHash3_x64_128(bytes, seed=42).h1 from the canonical C++ implementation (see make_golden.py)",
               "seed": 42, "vectors": vecs}, open(os.path.join(HERE, "murmur3_x64_128_vectors.json"), "w"), indent=0)

    ref_fa = "/root/reference/doc/example.fa"
    if os.path.exists(ref_fa):
        shutil.copyfile(ref_fa, os.path.join(HERE, "example.fa"))
    seq = read_fasta(os.path.join(HERE, "example.fa"))
    out = {"_comment": "derived by the validated restatement (oracle/kct_oracle.c), not by running the Rust reference",
           "length": len(seq), "k": {}}
    for k in (21, 31, 51):
        t = oracle.OracleTable(k)
        n = t.consume(seq)
        keys, counts = t.dump_arrays()
        hs, _ = oracle.seq_to_hashes(seq, k)
        tsv = "".join(f"{h}\t{c}\n" for h, c in zip(keys.tolist(), counts.tolist()))
        histo = {}
        for c in counts.tolist():
            histo[c] = histo.get(c, 0) + 1
        x = 0
        s = 0
        for h, c in zip(keys.tolist(), counts.tolist()):
            x ^= (h * c) & (2**64 - 1)
            s = (s + h * c) & (2**64 - 1)
        out["k"][str(k)] = {"n": n, "distinct": len(t), "max": int(counts.max()), "consumed": t.consumed,
                            "histo": {str(a): b for a, b in sorted(histo.items())},
                            "first3": [int(v) for v in hs[:3]], "min_hash": int(keys[0]),
                            "xor_hash_times_count": x, "sum_hash_times_count": s,
                            "sha256_dump_sortkeys_tsv": hashlib.sha256(tsv.encode()).hexdigest()}
    json.dump(out, open(os.path.join(HERE, "example_fa_digests.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
