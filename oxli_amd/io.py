"""Minimal FASTA / FASTQ record reader (plain or gzip) -- the role screed plays for the reference
(README.md:89-99).  Used by ``KmerCountTable.consume_file(skip_bad_kmers=False)`` and by tests; the
fast path parses in C++ inside libkct_hip.so (``kct_consume_file``)."""
import gzip

__all__ = ["read_records"]


def _open(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


def read_records(path):
    """Yields (name, sequence) with sequence as bytes (line breaks removed)."""
    with _open(path) as f:
        line = f.readline()
        while line and not line.strip():
            line = f.readline()
        if not line:
            return
        fmt = line[:1]
        if fmt not in (b">", b"@"):
            raise ValueError(f"{path}: neither FASTA nor FASTQ")
        while line:
            name = line[1:].strip().decode("utf-8", "replace")
            seq = []
            line = f.readline()
            if fmt == b">":
                while line and not line.startswith(b">"):
                    seq.append(line.rstrip(b"\r\n"))
                    line = f.readline()
                yield name, b"".join(seq)
            else:
                while line and not line.startswith(b"+"):
                    seq.append(line.rstrip(b"\r\n"))
                    line = f.readline()
                s = b"".join(seq)
                q = 0
                line = f.readline()
                while line and q < len(s):
                    q += len(line.rstrip(b"\r\n"))
                    line = f.readline()
                yield name, s
            while line and not line.strip():
                line = f.readline()
