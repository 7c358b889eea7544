"""seed -> chain hand-off (cs_chain_batch, host code of the library): the reference's mem_chain (mapping/comp_seed.cpp:241-285) restated,
against the chains the reference's own mem_chain built for the 14 golden runs (oracle/ref_harness.cpp --chains).  Input here are the
golden mems and seeds; tests/test_gpu_parity.py feeds the GPU engine's output through the same function."""
import os

import numpy as np
import pytest

import _data


def golden_chains(name, pname):
    return np.load(os.path.join(_data.GOLD, "%s.%s.chains.npz" % (name, pname)))


def check_chains(got, zc):
    assert np.array_equal(got["chain_off"], zc["chain_off"])
    ch = got["chains"]
    assert np.array_equal(ch["pos"], zc["pos"]) and np.array_equal(ch["rid"], zc["rid"]) and np.array_equal(ch["n_seeds"], zc["n"])
    assert np.array_equal(ch["frac_rep"].view(np.uint32), zc["frac_rep"].view(np.uint32)) and np.array_equal(ch["is_alt"], zc["is_alt"])
    cs = got["cseeds"]
    assert np.array_equal(cs["rbeg"], zc["seed_rbeg"]) and np.array_equal(cs["qbeg"], zc["seed_qbeg"]) and np.array_equal(cs["len"], zc["seed_len"])
    assert np.array_equal(np.diff(got["cseed_off"].astype(np.int64)), zc["n"])


@pytest.mark.parametrize("name,pname", _data.golden_runs())
@pytest.mark.parametrize("threads", [1, 3])
def test_chains_are_the_references(name, pname, threads):
    import compseed_amd as ca
    z, kw = _data.load_golden(name, pname)
    zc = golden_chains(name, pname)
    _, off = _data.load_reads(name)
    mems = np.zeros(z["mems"].shape[0], dtype=ca.INTV_DT)
    mems["x0"], mems["x1"], mems["x2"], mems["info"] = z["mems"][:, 0], z["mems"][:, 1], z["mems"][:, 2], z["mems"][:, 3]
    seeds = np.zeros(z["seed_rbeg"].size, dtype=ca.SEED_DT)
    seeds["rbeg"], seeds["qbeg"], seeds["len"] = z["seed_rbeg"], z["seed_qbeg"], z["seed_len"]
    c = ca.Chainer(_data.PREFIX)
    got = c.chain(z["mem_off"], mems, z["seed_off"], seeds, off, ca.ChainParams(k=kw.get("k", 19), c=kw.get("c", 500)), threads=threads)
    check_chains(got, zc)
    assert zc["pos"].size > 50
    c.close()


def test_goldens_exercise_the_tree():
    """the tandem-array reads: hundreds of chains per read (several levels of 9-key nodes) and reads whose chains share a key -- the cases
    in which a sorted array would not reproduce the B-tree's choice among equal keys"""
    zc = golden_chains("repeat100", "default")
    co = zc["chain_off"].astype(np.int64)
    per_read = np.diff(co)
    assert per_read.max() > 400
    dup = [r for r in range(per_read.size) if np.unique(zc["pos"][co[r]:co[r + 1]]).size < per_read[r]]
    assert len(dup) >= 3
    for r in range(per_read.size):
        assert (np.diff(zc["pos"][co[r]:co[r + 1]]) >= 0).all()     # traversal order is key order


def test_alt_contigs_are_flagged_like_the_reference(tmp_path):
    """an index with a <prefix>.alt file (hs38DH-style): the chainer reads it as bns_restore does (FM_index/bntseq.c:178-207) and the chains
    on the named contig carry is_alt = 1 (comp_seed.cpp:261); golden: the reference's own mem_chain run on such a prefix
    (tests/golden/make_golden.py alt).  The .alt fixture also has a header line, an unknown name and a last line without a newline."""
    import shutil
    import compseed_amd as ca
    alt_dir = os.path.join(os.path.dirname(_data.GOLD), "alt1")
    shutil.copy(_data.PREFIX + ".ann", tmp_path / "ref.ann")
    shutil.copy(os.path.join(alt_dir, "ref.alt"), tmp_path / "ref.alt")
    z, kw = _data.load_golden("main100", "default")
    zc = np.load(os.path.join(alt_dir, "main100.default.chains.npz"))
    _, off = _data.load_reads("main100")
    mems = np.zeros(z["mems"].shape[0], dtype=ca.INTV_DT)
    mems["x0"], mems["x1"], mems["x2"], mems["info"] = z["mems"][:, 0], z["mems"][:, 1], z["mems"][:, 2], z["mems"][:, 3]
    seeds = np.zeros(z["seed_rbeg"].size, dtype=ca.SEED_DT)
    seeds["rbeg"], seeds["qbeg"], seeds["len"] = z["seed_rbeg"], z["seed_qbeg"], z["seed_len"]
    c = ca.Chainer(str(tmp_path / "ref"))
    got = c.chain(z["mem_off"], mems, z["seed_off"], seeds, off, ca.ChainParams(), threads=2)
    check_chains(got, zc)
    assert set(zc["is_alt"].tolist()) == {0, 1}
    c.close()
