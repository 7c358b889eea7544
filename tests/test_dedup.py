"""cs_dedup_regions (dedup.cpp) = what the reference does with a read's alignment regions after the extension stage (comp_seed.cpp:2385-2395):
purged regions dropped, mem_sort_dedup_patch (comp_seed.cpp:629-687).  In: the regions the reference's extension stage left (tests/golden/aln1,
flt1, ddp1/gap3k.aln.npz); out must be what the reference's own mem_sort_dedup_patch left (tests/golden/ddp1/, oracle/_ref/ref_dump --dedup),
region by region in its order: two unstable sorts whose tie order is klib's, redundant regions removed, and -- gap3k: 3-kb reads with a
gap in the middle that the extension does not cross -- pairs of regions merged after a banded global alignment (ksw_global2's score).
Host code: runs without a GPU (an aligner created with device -1)."""
import os

import numpy as np
import pytest

import _data

G = os.path.dirname(_data.GOLD)
SETS = {"main100": ("aln1", _data.GOLD), "sorted150": ("aln1", _data.GOLD), "ragged": ("aln1", _data.GOLD), "repeat100": ("aln1", _data.GOLD),
        "indel150_400": ("aln1", os.path.join(G, "aln1")), "long90": ("flt1", os.path.join(G, "flt1")), "gap3k": ("ddp1", os.path.join(G, "ddp1"))}


def _load(name):
    import compseed_amd as ca
    adir, rdir = SETS[name]
    z = np.load(os.path.join(G, adir, name + ".aln.npz"))
    zd = np.load(os.path.join(G, "ddp1", name + ".ddp.npz"))
    raw = open(os.path.join(rdir, name + ".txt"), "rb").read()
    reads = raw.split(b"\n")[:-1] if raw.endswith(b"\n") else raw.split(b"\n")
    bases, off = _data.pack_reads(reads)
    regs = np.zeros(z["reg_rb"].size, dtype=ca.ALNREG_DT)
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep", "chain"):
        regs[f] = z["reg_" + f]
    return z["reg_off"], regs, bases, off, zd


@pytest.mark.parametrize("name", sorted(SETS))
def test_regions_after_dedup_are_the_references(name):
    import compseed_amd as ca
    reg_off, regs, bases, off, zd = _load(name)
    al = ca.Aligner(_data.PREFIX, -1)
    got = al.dedup_regions(reg_off, regs, bases, off)
    al.close()
    assert np.array_equal(got["reg_off"], zd["reg_off"]), name
    g = got["regs"]
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0"):
        assert np.array_equal(g[f], zd["reg_" + f]), (name, f, int((g[f] != zd["reg_" + f]).sum()))
    assert np.array_equal(g["frac_rep"].view(np.uint32), zd["reg_frac_rep"].view(np.uint32))
    assert np.array_equal(got["n_comp"], zd["reg_n_comp"])
    assert g.size < regs.size


def test_goldens_exercise_merges_and_ties():
    zd = np.load(os.path.join(G, "ddp1", "gap3k.ddp.npz"))
    assert (zd["reg_n_comp"] > 1).sum() > 100                 # regions the patch merged
    z = np.load(os.path.join(G, "ddp1", "repeat100.ddp.npz"))
    per_read = np.diff(z["reg_off"].astype(np.int64))
    assert per_read.max() > 100                                # hundreds of regions per read: equal scores and equal ends, the sorts' tie order decides


def test_host_only_aligner_refuses_to_extend():
    import compseed_amd as ca
    al = ca.Aligner(_data.PREFIX, -1)
    with pytest.raises(ca.CSError) as ei:
        al.extend_chains(np.zeros(1, np.uint64), np.zeros(0, ca.CHAIN_DT), np.zeros(1, np.uint64), np.zeros(0, ca.SEED_DT), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert ei.value.code == -4
    al.close()
