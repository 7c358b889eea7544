"""reordered-reads ingest (cs_reader_t): host code of the library, no GPU.  Chunks are cut as the reference cuts them (the first
even read count that reaches -K bases, main.cpp:54,437), FASTQ is sniffed from the first byte (main.cpp:399-406), gzip is transparent."""
import gzip
import os

import numpy as np
import pytest

import _data


def _lines(name):
    return open(os.path.join(_data.GOLD, name + ".txt"), "rb").read().split(b"\n")[:-1]


def _reference_chunks(reads, chunk_bases):
    out, cur, size = [], [], 0
    for r in reads:
        cur.append(r); size += len(r)
        if size >= chunk_bases and len(cur) % 2 == 0:
            out.append(cur); cur, size = [], 0
    if cur:
        out.append(cur)
    return out


@pytest.mark.parametrize("name,chunk", [("main100", 20000), ("ragged", 3000), ("sorted150", 10 ** 9), ("main100", 1)])
def test_chunks_like_the_reference(name, chunk):
    import compseed_amd as ca
    reads = _lines(name)
    want = _reference_chunks(reads, chunk)
    got = []
    rd = ca.Reader(os.path.join(_data.GOLD, name + ".txt"), chunk)
    for bases, off in rd:
        got.append([bytes(bases[int(off[i]):int(off[i + 1])]) for i in range(off.size - 1)])
    assert got == want
    rd.close()


def test_gzip_fastq_crlf_and_last_line_without_newline(tmp_path):
    import compseed_amd as ca
    reads = _lines("sorted150")[:101]
    fq = b"".join(b"@r%d extra\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads))
    p1 = tmp_path / "r.fq.gz"
    with gzip.open(p1, "wb") as f:
        f.write(fq)
    p2 = tmp_path / "r.txt"
    p2.write_bytes(b"\r\n".join(reads))            # CRLF, and no terminator after the last read
    for p in (p1, p2):
        got = []
        for bases, off in ca.Reader(str(p), 4000):
            assert (off.size - 1) % 2 == 0 or len(got) + off.size - 1 == len(reads)   # only the last chunk may be odd
            got += [bytes(bases[int(off[i]):int(off[i + 1])]) for i in range(off.size - 1)]
        assert got == reads


def test_blocks_larger_than_the_scan_buffer_and_errors(tmp_path):
    """a 150 MB file crosses the reader's 64-MB block boundary in the middle of lines; an over-long line is an error code"""
    import compseed_amd as ca
    rng = np.random.default_rng(3)
    n = 1_000_000
    arr = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, 150))]
    arr = np.concatenate([arr, np.full((n, 1), 10, np.uint8)], axis=1)
    p = tmp_path / "big.txt"
    arr.tofile(p)
    tot = 0
    for bases, off in ca.Reader(str(p), 30_000_000):
        k = off.size - 1
        assert (np.diff(off.astype(np.int64)) == 150).all()
        assert np.array_equal(bases.reshape(k, 150), arr[tot: tot + k, :150])
        tot += k
    assert tot == n
    bad = tmp_path / "bad.txt"
    bad.write_bytes(b"A" * 70000 + b"\n")
    with pytest.raises(ca.CSError) as ei:
        next(ca.Reader(str(bad), 1000))
    assert ei.value.code == -5
    with pytest.raises(ca.CSError):
        ca.Reader(str(tmp_path / "missing.txt"), 1000)
