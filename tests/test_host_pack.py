"""cs_pack_reads (host_pack.cpp): the records the host variants upload, against a numpy restatement of the record format
(smem_split.hpp pack_reads_kernel; the GPU test test_host_packed_reads_equal_device_packed compares the two on the device)."""
import numpy as np
import pytest

import _data


def _records_numpy(bases, off):
    lut = np.full(256, 4, dtype=np.uint8)
    lut[:4] = np.arange(4)
    for ch, c in zip("ACGT", range(4)):
        lut[ord(ch)] = c
        lut[ord(ch.lower())] = c
    n = off.size - 1
    rec = np.zeros(((int(off[-1]) >> 5) + n, 4), dtype=np.uint32)
    for r in range(n):
        rb, re = int(off[r]), int(off[r + 1])
        codes = lut[bases[rb:re]]
        first = (rb >> 5) + r
        nrec = (re >> 5) + r + 1 - first
        for k in range(nrec):
            c = codes[k * 32:(k + 1) * 32]
            b = 0
            bad = 0xffffffff
            for j, v in enumerate(c):
                if v < 4:
                    b |= int(v) << (2 * j)
                    bad &= ~(1 << j)
            rec[first + k] = (b & 0xffffffff, b >> 32, bad, 0)
    return rec


@pytest.mark.parametrize("scalar", [False, True])
def test_records_of_the_golden_read_sets(scalar):
    import compseed_amd as ca
    for name in ("ragged", "main100"):
        bases, off = _data.load_reads(name)
        n = min(400, off.size - 1)
        bases, off = bases[:int(off[n])], off[:n + 1]
        got = ca.pack_reads(bases, off, threads=3, scalar=scalar)
        assert np.array_equal(got, _records_numpy(bases, off)), name


def test_every_byte_value_and_every_length():
    import compseed_amd as ca
    rng = np.random.default_rng(5)
    lens = np.concatenate([np.arange(0, 200), rng.integers(0, 70, 300)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    bases = rng.integers(0, 256, int(off[-1]), dtype=np.uint8)                 # every byte value, most of them ambiguous
    acgt = np.frombuffer(b"ACGTacgt\x00\x01\x02\x03", dtype=np.uint8)
    m = rng.random(bases.size) < 0.8
    bases[m] = acgt[rng.integers(0, acgt.size, int(m.sum()))]
    want = _records_numpy(bases, off)
    for threads, scalar in ((1, False), (4, False), (2, True)):
        assert np.array_equal(ca.pack_reads(bases, off, threads=threads, scalar=scalar), want)
    # the vector path on a large input cut over many threads equals the scalar path
    off2 = np.arange(0, 151 * 60001, 151, dtype=np.uint64)
    b2 = acgt[rng.integers(0, acgt.size, int(off2[-1]))]
    b2[rng.integers(0, b2.size, 5000)] = ord("N")
    assert np.array_equal(ca.pack_reads(b2, off2, threads=8), ca.pack_reads(b2, off2, threads=1, scalar=True))


def test_bad_arguments_are_codes():
    import compseed_amd as ca
    with pytest.raises(ca.CSError):
        ca.pack_reads(np.zeros(10, np.uint8), np.array([0, 6, 4, 10], dtype=np.uint64))
    assert ca.pack_reads(np.zeros(0, np.uint8), np.array([0], dtype=np.uint64)).shape == (0, 4)
