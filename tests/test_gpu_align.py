"""The extension stage as a whole on the GPU box (cs_extend_chains, compseed_amd/csrc/align.cpp + extend.hip) against the REAL reference's
mem_chain2aln_across_reads_V2 (mapping/comp_seed.cpp:1319-2237): tests/golden/aln1/ holds, for five read sets, the chains the reference fed
into it (after its own mem_chain_flt / mem_flt_chained_seeds) and every alignment region it left -- rb re qb qe rid score truesc w seedcov
seedlen0 frac_rep, the region's chain, purged regions (qb = qe = -1) included (oracle/ref_harness.cpp --aln, tests/golden/make_golden.py aln).
The library must reproduce every field of every region, in the reference's order."""
import os

import numpy as np
import pytest

import _data

pytestmark = pytest.mark.gpu
ALN = os.path.join(os.path.dirname(_data.GOLD), "aln1")


FLT = os.path.join(os.path.dirname(_data.GOLD), "flt1")   # long reads (800-1500 bases): tests/golden/make_golden.py flt


DDP = os.path.join(os.path.dirname(_data.GOLD), "ddp1")   # regions after mem_sort_dedup_patch; gap3k: 3-kb reads with a gap in the middle (make_golden.py ddp)


def _load(name):
    z = np.load(os.path.join(FLT if name.startswith("long") else DDP if name.startswith("gap") else ALN, name + ".aln.npz"))
    rd_dir = ALN if name.startswith("indel") else FLT if name.startswith("long") else DDP if name.startswith("gap") else _data.GOLD
    raw = open(os.path.join(rd_dir, name + ".txt"), "rb").read()
    reads = raw.split(b"\n")[:-1] if raw.endswith(b"\n") else raw.split(b"\n")
    bases, off = _data.pack_reads(reads)
    return z, bases, off


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
@pytest.mark.parametrize("name", ["main100", "sorted150", "ragged", "repeat100", "indel150_400", "long90", "gap3k"])
def test_alignment_regions_are_the_references(name, flags):
    """flags: cs_aln_params_t.flags -- 0 = chains of up to 8 seeds a lane each / reads of up to 64 regions in registers, the rest a wave per chain / the
    LDS purge; 1 = everything through the wave-per-chain and LDS kernels; 2 = reads of more than 64 regions purged from HBM (the fallback beyond the LDS's
    3,000 regions); 3 = both.  Every path must give the reference's regions."""
    import compseed_amd as ca
    z, bases, off = _load(name)
    n_chains = z["chain_pos"].size
    chains = np.zeros(n_chains, dtype=ca.CHAIN_DT)
    chains["pos"], chains["rid"], chains["n_seeds"], chains["frac_rep"], chains["is_alt"] = z["chain_pos"], z["chain_rid"], z["chain_n"], z["chain_frac_rep"], z["chain_is_alt"]
    cseed_off = np.zeros(n_chains + 1, dtype=np.uint64); np.cumsum(z["chain_n"].astype(np.uint64), out=cseed_off[1:])
    cseeds = np.zeros(z["cseed_rbeg"].size, dtype=ca.SEED_DT)
    cseeds["rbeg"], cseeds["qbeg"], cseeds["len"] = z["cseed_rbeg"], z["cseed_qbeg"], z["cseed_len"]
    al = ca.Aligner(_data.PREFIX, 0, ca.AlnParams(flags=flags))
    got = al.extend_chains(z["chain_off"], chains, cseed_off, cseeds, bases, off, cseed_score=z["cseed_score"])
    st = al.stats()
    per_read = np.diff(got["reg_off"].astype(np.int64))
    if name == "repeat100": assert per_read.max() > 64 and z["chain_n"].max() > 8   # (the heavy paths are exercised at flags 0, too)
    al.close()
    assert np.array_equal(got["reg_off"], z["reg_off"])
    g = got["regs"]
    purged = (z["reg_qb"] == -1) & (z["reg_qe"] == -1)
    assert np.array_equal((g["qb"] == -1) & (g["qe"] == -1), purged), (name, int(((g["qb"] == -1) & (g["qe"] == -1)).sum()), int(purged.sum()))
    for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "chain"):
        assert np.array_equal(g[f], z["reg_" + f]), (name, f, int((g[f] != z["reg_" + f]).sum()))
    assert np.array_equal(g["frac_rep"].view(np.uint32), z["reg_frac_rep"].view(np.uint32))
    assert st["regions"] == g.size and st["purged"] == int(purged.sum()) and st["pairs"] >= st["regions"] // 2
    assert g.size > 4000 and purged.sum() > 1000


@pytest.mark.parametrize("name", ["sorted150", "indel150_400", "long90", "gap3k"])
def test_reads_to_regions_through_the_abi_equal_the_references(name):
    """the library's whole side of comp_seed.cpp:2242-2395 from the reads: GPU seeding -> cs_chain_batch -> cs_chain_filter ->
    cs_extend_chains -> cs_dedup_regions; the regions are the reference's, field by field, after the extension stage and after
    mem_sort_dedup_patch"""
    import compseed_amd as ca
    z, bases, off = _load(name)
    ix = ca.Index.load(_data.PREFIX)
    eng = ca.Engine(ix, 0)
    res = eng.seed_batch(bases, off, ca.Params())
    ch = ca.Chainer(_data.PREFIX)
    c = ch.chain(res.mem_off, res.mems, res.seed_off, res.seeds, off, ca.ChainParams(), threads=2)
    f = ch.filter(c["chain_off"], c["chains"], c["cseed_off"], c["cseeds"], bases, off, threads=2)
    assert np.array_equal(f["chain_off"], z["chain_off"]) and np.array_equal(f["chains"]["pos"], z["chain_pos"]) and np.array_equal(f["cseed_score"], z["cseed_score"])
    al = ca.Aligner(_data.PREFIX, 0)
    got = al.extend_chains(f["chain_off"], f["chains"], f["cseed_off"], f["cseeds"], bases, off, cseed_score=f["cseed_score"])
    g = got["regs"]
    assert np.array_equal(got["reg_off"], z["reg_off"])
    for fld in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "chain"):
        assert np.array_equal(g[fld], z["reg_" + fld]), (name, fld)
    zd = np.load(os.path.join(DDP, name + ".ddp.npz"))
    dd = al.dedup_regions(got["reg_off"], g, bases, off)
    assert np.array_equal(dd["reg_off"], zd["reg_off"]) and np.array_equal(dd["n_comp"], zd["reg_n_comp"])
    for fld in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0"):
        assert np.array_equal(dd["regs"][fld], zd["reg_" + fld]), (name, fld)
    al.close(); ch.close(); eng.close(); ix.close()


def test_engine_chainer_aligner_end_to_end():
    """reads -> GPU seeding -> chains (cs_chain_batch) -> GPU extension, all through the C ABI: the chains are unfiltered here (the reference's
    chain filters are the caller's), so the regions are compared with what the same driver gives for the golden's own seeds -- and every
    surviving region must be a consistent local alignment frame (inside the read and the reference, score at least the seed's)"""
    import compseed_amd as ca
    bases, off = _data.load_reads("sorted150")
    ix = ca.Index.load(_data.PREFIX)
    eng = ca.Engine(ix, 0)
    res = eng.seed_batch(bases, off, ca.Params())
    ch = ca.Chainer(_data.PREFIX)
    c = ch.chain(res.mem_off, res.mems, res.seed_off, res.seeds, off, ca.ChainParams(), threads=2)
    al = ca.Aligner(_data.PREFIX, 0)
    got = al.extend_chains(c["chain_off"], c["chains"], c["cseed_off"], c["cseeds"], bases, off)
    g = got["regs"]
    assert g.size == c["cseeds"].size
    live = ~((g["qb"] == -1) & (g["qe"] == -1))
    read_of = np.repeat(np.arange(off.size - 1), np.diff(got["reg_off"].astype(np.int64)))
    rl = np.diff(off.astype(np.int64))[read_of]
    assert (g["qb"][live] >= 0).all() and (g["qe"][live] <= rl[live]).all() and (g["qb"][live] < g["qe"][live]).all()
    assert (g["rb"][live] >= 0).all() and (g["re"][live] > g["rb"][live]).all() and (g["score"][live] >= g["seedlen0"][live]).all()
    assert live.sum() > 1000
    al.close(); ch.close(); eng.close(); ix.close()


def test_degenerate_inputs():
    """no reads; reads without chains; a chain the seed test left without seeds: empty or pass-through results, error codes for bad CSR"""
    import compseed_amd as ca
    al = ca.Aligner(_data.PREFIX, 0)
    z = np.zeros
    got = al.extend_chains(z(1, np.uint64), z(0, ca.CHAIN_DT), z(1, np.uint64), z(0, ca.SEED_DT), z(0, np.uint8), z(1, np.uint64))
    assert got["regs"].size == 0 and np.array_equal(got["reg_off"], [0])
    bases, off = _data.load_reads("sorted150")
    bases, off = bases[:int(off[3])], off[:4]
    got = al.extend_chains(z(4, np.uint64), z(0, ca.CHAIN_DT), z(1, np.uint64), z(0, ca.SEED_DT), bases, off)      # three reads, no chains
    assert got["regs"].size == 0 and np.array_equal(got["reg_off"], [0, 0, 0, 0])
    zz, _, _ = _load("sorted150")
    # the first read's first chain, once as it is and once emptied: the emptied one contributes nothing
    n0 = int(zz["chain_n"][0])
    chains = np.zeros(2, dtype=ca.CHAIN_DT)
    for f, k in (("pos", "chain_pos"), ("rid", "chain_rid"), ("frac_rep", "chain_frac_rep"), ("is_alt", "chain_is_alt")):
        chains[f] = zz[k][0]
    chains["n_seeds"] = [n0, 0]
    cseeds = np.zeros(n0, dtype=ca.SEED_DT)
    cseeds["rbeg"], cseeds["qbeg"], cseeds["len"] = zz["cseed_rbeg"][:n0], zz["cseed_qbeg"][:n0], zz["cseed_len"][:n0]
    b1, o1 = bases[:int(off[1])], off[:2]
    got = al.extend_chains(np.array([0, 2], np.uint64), chains, np.array([0, n0, n0], np.uint64), cseeds, b1, o1, cseed_score=zz["cseed_score"][:n0])
    assert got["regs"].size == n0 and np.array_equal(got["regs"]["score"], zz["reg_score"][:n0])
    with pytest.raises(ca.CSError):
        al.extend_chains(np.array([0, 2], np.uint64), chains, np.array([0, n0 - 1, n0], np.uint64), cseeds, b1, o1)   # offsets that do not match the seed counts
    al.close()
