"""GPU index builder (cs_index_build) against the reference-built fixture: .bwt/.sa must be byte-identical."""
import hashlib
import json
import os

import numpy as np
import pytest

import _data

pytestmark = pytest.mark.gpu


def test_pac_roundtrip_matches_fasta():
    import gzip
    fa = gzip.open(os.path.join(_data.GOLD, "ref.fa.gz")).read().decode().split("\n")
    g = "".join(l for l in fa if not l.startswith(">"))
    codes = _data.load_pac_forward()
    assert codes.size == len(g)
    lut = {"A": 0, "C": 1, "G": 2, "T": 3}
    want = np.array([lut.get(c, -1) for c in g])
    known = want >= 0
    assert np.array_equal(codes[known], want[known])  # N positions hold lrand48 bases (bntseq.c:295)


def test_built_index_is_byte_identical_to_bwaidx(tmp_path):
    import compseed_amd as ca
    fwd = _data.load_pac_forward()
    ix = ca.Index.build(fwd, 0)
    f = _data.load_bwt_files()
    assert ix.view.primary == f["primary"]
    assert list(ix.view.L2)[1:] == [int(x) for x in f["L2"]]
    bwt, sa = ix.arrays()
    assert np.array_equal(bwt, f["bwt"])
    assert np.array_equal(sa, f["sa"])
    ix.save(str(tmp_path / "mine"))
    man = json.load(open(os.path.join(os.path.dirname(_data.GOLD), "MANIFEST.json")))["md5"]
    for ext in ("bwt", "sa"):
        assert hashlib.md5(open(tmp_path / ("mine." + ext), "rb").read()).hexdigest() == man["g1/ref." + ext]
    ix.close()


def test_builder_on_adversarial_texts():
    """long runs, tandem repeats and tiny genomes: the doubling rounds must converge and agree with a naive suffix sort"""
    import compseed_amd as ca
    rng = np.random.default_rng(3)
    cases = [np.zeros(1, np.uint8), np.array([0, 1, 2, 3], np.uint8), np.zeros(500, np.uint8),
             np.tile(np.array([0, 1], np.uint8), 300), np.tile(rng.integers(0, 4, 37).astype(np.uint8), 40),
             rng.integers(0, 4, 5000).astype(np.uint8),
             np.concatenate([rng.integers(0, 4, 700), np.full(900, 3), rng.integers(0, 4, 300)]).astype(np.uint8)]
    for g in cases:
        T = np.concatenate([g, 3 - g[::-1]])
        n = T.size
        s = T.tobytes()
        sa = sorted(range(n + 1), key=lambda i: s[i:])           # $ = end of string sorts first
        primary = sa.index(0)
        bw = [T[i - 1] for i in sa if i != 0]
        ix = ca.Index.build(g, 0)
        assert ix.view.primary == primary and ix.view.seq_len == n
        words, samp = ix.arrays()
        got = []
        for b in range((n + 127) // 128):
            for w in range(8):
                for t in range(16):
                    pos = b * 128 + w * 16 + t
                    if pos < n:
                        got.append((int(words[b * 16 + 8 + w]) >> ((15 - t) * 2)) & 3)
        assert got == [int(x) for x in bw]
        assert [int(x) for x in samp[1:]] == [sa[r] for r in range(32, n + 1, 32)]
        ix.close()


def test_64bit_builder_instantiation(tmp_path):
    """genomes beyond 2^31 bp (hg19) use 64-bit suffix indices; the same code path must reproduce the bwaidx fixture"""
    import compseed_amd as ca
    ix = ca.Index.build(_data.load_pac_forward(), 0, force_64bit=True)
    f = _data.load_bwt_files()
    bwt, sa = ix.arrays()
    assert ix.view.primary == f["primary"] and np.array_equal(bwt, f["bwt"]) and np.array_equal(sa, f["sa"])
    ix.close()


def test_index_from_fasta_writes_all_five_files_like_bwaidx(tmp_path):
    """cs_index_build_fasta == bwa_idx_build (index_main.c:257-325): .bwt .sa .pac .ann .amb, md5-identical to the reference's bwaidx output"""
    import hashlib
    import json
    import compseed_amd as ca
    man = json.load(open(os.path.join(_data.HERE, "golden", "MANIFEST.json")))["md5"]
    ca.build_index_from_fasta(os.path.join(_data.GOLD, "ref.fa.gz"), str(tmp_path / "idx"), 0)
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert hashlib.md5(open(tmp_path / ("idx." + ext), "rb").read()).hexdigest() == man["g1/ref." + ext], ext


def test_index_from_fasta_at_bwaidx_large_genome_branch(tmp_path):
    """64 Mbp in three contigs (tests/_data.bigref_fasta): above 50 M packed bases the reference's bwaidx builds the BWT with bwt_bwtgen2 instead of
    the in-memory sort (FM_index/index_main.c:277-283).  Its five files' md5s are in tests/golden/c2/config2.json (make_golden.py bigref); the GPU
    builder must reproduce every one of them -- and the engine's own check of the result must be clean."""
    import compseed_amd as ca
    gold = json.load(open(os.path.join(_data.HERE, "golden", "c2", "config2.json")))
    fa = str(tmp_path / "big.fa")
    assert _data.bigref_fasta(fa) == gold["bases"]
    assert hashlib.md5(open(fa, "rb").read()).hexdigest() == gold["fasta_md5"]            # the same genome as the one bwaidx indexed
    ca.build_index_from_fasta(fa, str(tmp_path / "big"), 0)
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert hashlib.md5(open(tmp_path / ("big." + ext), "rb").read()).hexdigest() == gold["index_md5"][ext], ext
    ix = ca.Index.load(str(tmp_path / "big"))
    eng = ca.Engine(ix, 0)
    chk = eng.check_index()                                                                # (no genome on the device here: everything but the text comparison)
    assert chk["rows_checked"] == 2 * gold["bases"] and chk["text_checked"] == 0
    assert all(chk[k] == 0 for k in ("order_violations", "isa_violations", "bwt_violations", "sampled_sa_violations", "undecided_rows")), chk
    eng.close(); ix.close()


def test_index_check_notices_a_corrupted_index():
    """the checker is not vacuous: swap two sampled SA entries / flip BWT bases of the fixture and the counts are non-zero"""
    import compseed_amd as ca
    f = _data.load_bwt_files()
    good = ca.Index.from_arrays(f["primary"], f["L2"], f["bwt"], f["sa"], f["sa_intv"])
    e = ca.Engine(good, 0)
    c = e.check_index()
    assert all(c[k] == 0 for k in ("order_violations", "isa_violations", "bwt_violations", "sampled_sa_violations", "undecided_rows"))
    e.close(); good.close()
    sa = f["sa"].copy(); sa[[100, 2000]] = sa[[2000, 100]]
    bad = ca.Index.from_arrays(f["primary"], f["L2"], f["bwt"], sa, f["sa_intv"])
    e = ca.Engine(bad, 0)
    c = e.check_index()
    assert c["order_violations"] + c["isa_violations"] + c["bwt_violations"] + c["sampled_sa_violations"] > 0
    e.close(); bad.close()
