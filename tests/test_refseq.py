"""FASTA front end and .pac / .ann / .amb writers (host code of the library, no GPU): byte-identical to the files the reference's own
bwaidx wrote for the fixture genome (two contigs, an N run replaced through srand48(11) / lrand48, FM_index/bntseq.c:266,295)."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

import _data


def _md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def test_fasta_to_pac_ann_amb_is_bwaidx_identical(tmp_path):
    import compseed_amd as ca
    man = json.load(open(os.path.join(_data.HERE, "golden", "MANIFEST.json")))["md5"]
    r = ca.RefSeq(os.path.join(_data.GOLD, "ref.fa.gz"))
    assert r.n_seqs == 2 and r.n_holes == 1 and r.l_pac == _data.load_pac_forward().size
    assert np.array_equal(r.codes, _data.load_pac_forward())          # incl. the 57 bases drawn for the N run
    r.save(str(tmp_path / "x"))
    for ext in ("pac", "ann", "amb"):
        assert _md5(tmp_path / ("x." + ext)) == man["g1/ref." + ext], ext
        assert open(tmp_path / ("x." + ext), "rb").read() == open(os.path.join(_data.GOLD, "ref." + ext), "rb").read()
    r.close()


def test_fasta_details(tmp_path):
    """header comments, lower case, IUPAC codes (a run of one code is one hole, a change of code opens a new one), blank lines, CRLF,
    FASTQ records, a sequence length that is a multiple of 4 (the extra zero byte of the .pac), plain (not gzip) input"""
    import compseed_amd as ca
    fa = tmp_path / "t.fa"
    fa.write_bytes(b">c1 first contig\r\nACGTacgtNNNRRNAC\r\n\r\nGT\n>c2\nNACGTYYKACGTACGA\n@q1 a read\nACGTN\n+\nIIIII\n")
    r = ca.RefSeq(str(fa))
    assert (r.n_seqs, r.l_pac) == (3, 18 + 16 + 5)
    r.save(str(tmp_path / "t"))
    ann = open(tmp_path / "t.ann").read().splitlines()
    assert ann[0] == "39 3 11" and ann[1] == "0 c1 first contig" and ann[2] == "0 18 3" and ann[3] == "0 c2 (null)" and ann[4] == "18 16 3"
    assert ann[5] == "0 q1 a read" and ann[6] == "34 5 1"
    amb = open(tmp_path / "t.amb").read().splitlines()
    assert amb == ["39 3 7", "8 3 N", "11 2 R", "13 1 N", "18 1 N", "23 2 Y", "25 1 K", "38 1 N"]
    pac = np.fromfile(tmp_path / "t.pac", dtype=np.uint8)
    assert pac.size == 39 // 4 + 1 + 1 and pac[-1] == 3
    keep = np.array([i for i in range(39) if i not in (8, 9, 10, 11, 12, 13, 18, 23, 24, 25, 38)])
    want = _data.codes_of("ACGTACGTNNNNNNACGTNACGTNNNACGTACGAACGTN".upper())
    assert np.array_equal(r.codes[keep], want[keep]) and (r.codes <= 3).all()
    fa4 = tmp_path / "m4.fa"
    fa4.write_text(">x\nACGTACGT\n")
    r4 = ca.RefSeq(str(fa4)); r4.save(str(tmp_path / "m4"))
    assert list(np.fromfile(tmp_path / "m4.pac", dtype=np.uint8)) == [0x1B, 0x1B, 0, 0]
    with pytest.raises(ca.CSError):
        ca.RefSeq(str(tmp_path / "missing.fa"))
    r.close(); r4.close()
