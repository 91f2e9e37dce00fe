"""CPU test of the Python-side FASTA/FASTQ reader (the role screed plays for the reference)."""
import gzip
import os

from oxli_amd.io import read_records

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_example_fa_is_one_record():
    recs = list(read_records(os.path.join(GOLDEN, "example.fa")))
    assert len(recs) == 1
    name, seq = recs[0]
    assert name.startswith("CP001071.1") and len(seq) == 349930 and set(seq) <= set(b"ACGT")


def test_fasta_fastq_gzip(tmp_path):
    fa = tmp_path / "x.fa"
    fa.write_text(">a desc\nACGT\nAC\n\n>b\n>c\r\nGG\r\nTT\r\n")
    assert list(read_records(str(fa))) == [("a desc", b"ACGTAC"), ("b", b""), ("c", b"GGTT")]
    fq = tmp_path / "x.fq.gz"
    with gzip.open(fq, "wt") as f:
        f.write("@r1\nACGTN\n+\nIIIII\n@r2\nGG\n+r2\n@@\n")
    assert list(read_records(str(fq))) == [("r1", b"ACGTN"), ("r2", b"GG")]
    empty = tmp_path / "e.fa"
    empty.write_text("\n")
    assert list(read_records(str(empty))) == []
