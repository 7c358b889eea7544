"""cs_chain_filter (chain_filter.cpp) = the reference's mem_chain_flt + mem_flt_chained_seeds (mapping/comp_seed.cpp:297-412), host code:
the unfiltered golden chains (oracle/ref_harness.cpp --chains: what the reference's mem_chain built) go in, and what must come out are
the chains the reference handed to its extension stage (tests/golden/aln1/, oracle/_ref/ref_dump --aln: mem_chain -> mem_chain_flt ->
mem_flt_chained_seeds) -- chain by chain in the reference's order (weights tie often: the sort is klib's introsort, restated), seed by
seed, score by score.  tests/golden/flt1/ holds reads long enough for the seed test (mem_seed_sw, ksw_align2's score)."""
import os

import numpy as np
import pytest

import _data
from test_chain import golden_chains

ALN = os.path.join(os.path.dirname(_data.GOLD), "aln1")


def _chains_in(zc):
    import compseed_amd as ca
    chains = np.zeros(zc["pos"].size, dtype=ca.CHAIN_DT)
    chains["pos"], chains["rid"], chains["n_seeds"], chains["frac_rep"], chains["is_alt"] = zc["pos"], zc["rid"], zc["n"], zc["frac_rep"], zc["is_alt"]
    cseed_off = np.zeros(zc["pos"].size + 1, dtype=np.uint64); np.cumsum(zc["n"].astype(np.uint64), out=cseed_off[1:])
    cseeds = np.zeros(zc["seed_rbeg"].size, dtype=ca.SEED_DT)
    cseeds["rbeg"], cseeds["qbeg"], cseeds["len"] = zc["seed_rbeg"], zc["seed_qbeg"], zc["seed_len"]
    return zc["chain_off"], chains, cseed_off, cseeds


def check_filtered(got, z):
    assert np.array_equal(got["chain_off"], z["chain_off"])
    ch = got["chains"]
    assert np.array_equal(ch["pos"], z["chain_pos"]) and np.array_equal(ch["rid"], z["chain_rid"]) and np.array_equal(ch["n_seeds"], z["chain_n"])
    assert np.array_equal(ch["frac_rep"].view(np.uint32), z["chain_frac_rep"].view(np.uint32)) and np.array_equal(ch["is_alt"], z["chain_is_alt"])
    cs = got["cseeds"]
    assert np.array_equal(cs["rbeg"], z["cseed_rbeg"]) and np.array_equal(cs["qbeg"], z["cseed_qbeg"]) and np.array_equal(cs["len"], z["cseed_len"])
    assert np.array_equal(got["cseed_score"], z["cseed_score"])


@pytest.mark.parametrize("name", ["main100", "repeat100", "sorted150", "ragged"])
@pytest.mark.parametrize("threads", [1, 3])
def test_filtered_chains_are_the_references(name, threads):
    import compseed_amd as ca
    zc = golden_chains(name, "default")
    z = np.load(os.path.join(ALN, name + ".aln.npz"))
    bases, off = _data.load_reads(name)
    c = ca.Chainer(_data.PREFIX)
    got = c.filter(*_chains_in(zc), bases, off, threads=threads)
    check_filtered(got, z)
    assert got["chains"].size < zc["pos"].size            # the filter dropped something
    c.close()


def test_goldens_exercise_the_tie_order():
    """reads with hundreds of chains of equal weight (the tandem-array reads): all of them survive, in the order klib's introsort leaves
    them in -- neither the input order nor a stable sort's"""
    zc = golden_chains("repeat100", "default")
    z = np.load(os.path.join(ALN, "repeat100.aln.npz"))
    per_read = np.diff(zc["chain_off"].astype(np.int64))
    kept = np.diff(z["chain_off"].astype(np.int64))
    assert per_read.max() > 100 and kept.max() > 100         # more than 16: the quicksort part of the introsort runs, not only its insertion sort
    shuffled = 0
    for r in np.nonzero(kept > 100)[0]:
        a = zc["pos"][zc["chain_off"][r]:zc["chain_off"][r + 1]]
        b = z["chain_pos"][z["chain_off"][r]:z["chain_off"][r + 1]]
        if a.size == b.size and sorted(a.tolist()) == sorted(b.tolist()) and not np.array_equal(a, b):
            shuffled += 1
    assert shuffled > 10


FLT = os.path.join(os.path.dirname(_data.GOLD), "flt1")


def _long_reads():
    reads = [l.encode() for l in open(os.path.join(FLT, "long90.txt")).read().split("\n") if l]
    return _data.pack_reads(reads)


@pytest.mark.parametrize("threads", [1, 4])
def test_long_reads_seed_test_is_the_references(threads):
    """reads of 800-1500 bases: mem_flt_chained_seeds runs (5.5 ln(l) <= 0.05 l), every short seed is scored by a local alignment of its
    neighbourhood -- ksw_align2's number -- and dropped below ~5.5 ln(l): the surviving seeds and their scores are the reference's"""
    import compseed_amd as ca
    zc = np.load(os.path.join(FLT, "long90.chains.npz"))
    z = np.load(os.path.join(FLT, "long90.aln.npz"))
    bases, off = _long_reads()
    c = ca.Chainer(_data.PREFIX)
    got = c.filter(*_chains_in(zc), bases, off, threads=threads)
    check_filtered(got, z)
    assert (z["cseed_score"] != z["cseed_len"]).sum() > 1000           # scores that are alignment scores, not seed lengths
    assert zc["seed_rbeg"].size - z["cseed_rbeg"].size > 1000          # seeds dropped by the chain filter and by the seed test
    with pytest.raises(ca.CSError):
        c.filter(*_chains_in(zc), None, off)                           # the seed test needs the reads
    c.close()
