#!/usr/bin/env python3
"""Hand-assembles tests/golden/reference_format_save.json.gz: a file as the REFERENCE's `save()` writes it (lib.rs:274-292), built from
the format's description -- not by this repository's writer -- so that `load()` is pinned to the reference's wire format:

  * the text is `serde_json::to_string(&self)` of the struct at lib.rs:31-39: the fields in declaration order (counts, ksize, version,
    consumed, store_kmers, hash_to_kmer), no whitespace, u64 map keys as JSON strings, `Option::None` as null; a `HashMap` iterates in
    no particular order, so the keys below are deliberately NOT sorted;
  * the container is gzip written by niffler / flate2 at `Level::One`: a 10-byte header with no name and no time stamp
    (1f 8b 08 00 00000000, XFL = 4 "fastest", OS = 255 "unknown"), one deflate stream, CRC-32 and ISIZE.

The hashes are the reference's own known answers (tests/golden/reference_kats.json: k = 4 k-mers from
/root/reference/src/python/tests/test_kmers_and_hashes.py:13-16, 43-45 and test_dump.py:52-54); the counts are arbitrary.

    python tests/golden/make_reference_save_fixture.py        (no GPU, no reference checkout needed)
"""
import os
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
PAIRS = [   # (canonical 4-mer, reference hash, count) in "HashMap order": unsorted
    ("AACC", 6779379503393060785, 3),
    ("AAAA", 17832910516274425539, 1),
    ("ACGT", 2597925387403686983, 18446744073709551615),   # u64::MAX: serde_json writes integers in full
    ("ATAA", 179996601836427478, 2),
    ("CCCC", 73459868045630124, 7),
    ("AACG", 7952982457453691616, 40),
    ("AAAC", 9097280691811734508, 5),
]
CONSUMED = 123456789012


def text(store_kmers, version="0.3.0"):
    counts = ",".join(f'"{h}":{c}' for _k, h, c in PAIRS)
    h2k = "{" + ",".join(f'"{h}":"{k}"' for k, h, _c in reversed(PAIRS)) + "}" if store_kmers else "null"
    return (f'{{"counts":{{{counts}}},"ksize":4,"version":"{version}","consumed":{CONSUMED},'
            f'"store_kmers":{"true" if store_kmers else "false"},"hash_to_kmer":{h2k}}}')


def gzip_level_one(data):
    deflater = zlib.compressobj(1, zlib.DEFLATED, -15)
    body = deflater.compress(data) + deflater.flush()
    return b"\x1f\x8b\x08\x00" + struct.pack("<I", 0) + b"\x04\xff" + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def main():
    with open(os.path.join(HERE, "reference_format_save.json.gz"), "wb") as f:
        f.write(gzip_level_one(text(True).encode()))
    with open(os.path.join(HERE, "reference_format_save_old_version.json"), "w") as f:   # niffler also reads uncompressed input
        f.write(text(False, version="0.2.9"))


if __name__ == "__main__":
    main()
