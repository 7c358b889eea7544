"""GPU parity tests: the HIP engine, called through the C ABI, against the golden vectors of the real reference
and against the oracle (CPU restatement) on the same inputs.  Bit-exact: these are integers."""
import os

import numpy as np
import pytest

import _data
import _oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0)
    yield e
    e.close()
    ix.close()


def test_primitives_known_answers(eng):
    import compseed_amd as ca
    z = np.load(os.path.join(_data.GOLD, "prims.npz"))
    occ = z["occ4"]
    assert np.array_equal(eng.occ4(occ[:, 0]), occ[:, 1:5])          # bwt_occ4, bwt.c:169
    ext = z["ext"]
    ik = np.zeros(ext.shape[0], dtype=ca.INTV_DT)
    ik["x0"], ik["x1"], ik["x2"] = ext[:, 0], ext[:, 1], ext[:, 2]
    got = eng.extend(ik, ext[:, 3].astype(np.uint8))                    # bwt_extend, bwt.c:262
    want = ext[:, 4:].reshape(-1, 4, 3)
    assert np.array_equal(got["x0"], want[:, :, 0])
    assert np.array_equal(got["x1"], want[:, :, 1])
    assert np.array_equal(got["x2"], want[:, :, 2])
    assert not got["info"].any()                                        # single-child path == four-child path
    sa = z["sa"]
    assert np.array_equal(eng.sa(sa[:, 0]), sa[:, 1])                   # bwt_sa, bwt.c:86


def _check_against_golden(res, z):
    assert np.array_equal(res.mem_off, z["mem_off"])
    m = res.mems
    assert np.array_equal(np.stack([m["x0"], m["x1"], m["x2"], m["info"]], axis=1), z["mems"])
    assert np.array_equal(res.seed_off, z["seed_off"])
    assert np.array_equal(res.seeds["rbeg"], z["seed_rbeg"])
    assert np.array_equal(res.seeds["qbeg"], z["seed_qbeg"])
    assert np.array_equal(res.seeds["len"], z["seed_len"])


@pytest.mark.parametrize("name,pname", _data.golden_runs())
def test_golden_seeds(eng, name, pname):
    import compseed_amd as ca
    z, kw = _data.load_golden(name, pname)
    bases, off = _data.load_reads(name)
    eng.reset_stats()
    res = eng.seed_batch(bases, off, ca.Params(**kw))
    _check_against_golden(res, z)
    st = eng.stats()
    if st["overflow_reads"] == 0:                       # (reads that overflow into the second pass are seeded twice)
        # the default configuration skips extensions the reference performs (text shortcuts, window scheme), so its count
        # is only bounded by the reference's; sst_mode=0 reproduces it exactly (test_device_sst_is_transparent)
        assert st["bwt_calls"] <= int(z["counters"][3])
    assert st["sal_queries"] == int(z["counters"][5])


def test_device_variant_equals_host_variant(eng):
    import compseed_amd as ca
    bases, off = _data.load_reads("sorted150")
    host = eng.seed_batch(bases, off)
    d_b = eng.alloc(bases.nbytes + 64); d_o = eng.alloc(off.nbytes)
    eng.upload(d_b, bases); eng.upload(d_o, off)
    dev = eng.seed_batch_device(d_b, d_o, off.size - 1, bases.size)
    assert dev.n_mems == host.n_mems and dev.n_seeds == host.n_seeds
    mo = eng.download(dev.ptr["mem_off"], np.uint64, off.size)
    mm = eng.download(dev.ptr["mems"], ca.INTV_DT, dev.n_mems)
    ss = eng.download(dev.ptr["seeds"], ca.SEED_DT, dev.n_seeds)
    assert np.array_equal(mo, host.mem_off) and np.array_equal(mm, host.mems) and np.array_equal(ss, host.seeds)
    # the caller's read buffer is not modified (the reference overwrites it in place, comp_seed.cpp:2258)
    assert np.array_equal(eng.download(d_b, np.uint8, bases.size), bases)
    eng.free(d_b); eng.free(d_o)


def test_empty_and_degenerate_batches(eng):
    r = eng.seed_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert r.n_reads == 0 and r.n_mems == 0 and r.n_seeds == 0
    bases, off = _data.pack_reads([b"", b"", b""])
    r = eng.seed_batch(bases, off)
    assert r.n_mems == 0 and list(r.mem_off) == [0, 0, 0, 0]
    bases, off = _data.pack_reads([b"N" * 100, b"ACGT"])
    r = eng.seed_batch(bases, off)
    assert r.n_mems == 0


def test_errors_are_codes_not_aborts(eng):
    import compseed_amd as ca
    bases, off = _data.pack_reads([b"A" * 70000])
    with pytest.raises(ca.CSError) as ei:
        eng.seed_batch(bases, off)
    assert ei.value.code == -5  # CS_ERANGE: MAX_READ_LEN 65535 (main.cpp:83-86 aborts there)
    with pytest.raises(ca.CSError):
        eng.seed_batch(np.zeros(10, np.uint8), np.array([0, 8, 4], np.uint64))


def test_matches_oracle_on_fresh_random_reads(eng):
    """reads never seen by the golden generator: substrings of the fixture genome with heavy mutation + junk"""
    rng = np.random.default_rng(5)
    import gzip
    fa = gzip.open(os.path.join(_data.GOLD, "ref.fa.gz")).read().decode().split("\n")
    g = "".join(l for l in fa if not l.startswith(">")).replace("N", "A")
    reads = []
    for _ in range(4000):
        ln = int(rng.integers(20, 200)); p = int(rng.integers(0, len(g) - ln))
        r = np.frombuffer(g[p:p + ln].encode(), dtype=np.uint8).copy()
        mut = rng.random(ln) < rng.choice([0.0, 0.01, 0.03, 0.1])
        r[mut] = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, int(mut.sum()))]
        reads.append(r.tobytes())
    bases, off = _data.pack_reads(reads)
    o = _oracle.OracleIndex(_data.PREFIX)
    for kw in (dict(), dict(k=15, r=1.0, y=30, c=100, s=30)):
        want = o.seed_batch(bases, off, _oracle.make_params(**kw), mode=0, threads=4)
        import compseed_amd as ca
        got = eng.seed_batch(bases, off, ca.Params(**kw))
        assert np.array_equal(got.mem_off, want["mem_off"]) and np.array_equal(got.mems, want["mems"])
        assert np.array_equal(got.seed_off, want["seed_off"]) and np.array_equal(got.seeds, want["seeds"])
    o.close()


def test_full_sa_agrees_with_walk_everywhere(eng):
    """every row: the HBM-resident full suffix array == bwt_sa walked from the samples (sa_kernel poisons mismatches)"""
    n = int(eng._index.view.seq_len)
    rows = np.arange(0, n + 1, dtype=np.uint64)
    got = eng.sa(rows)
    assert not (got == np.uint64(0xdeadbeefdeadbeef)).any()
    o = _oracle.OracleIndex(_data.PREFIX)
    pick = np.random.default_rng(1).integers(1, n + 1, 3000)
    assert [int(got[k]) for k in pick] == [o.sa(int(k)) for k in pick]
    o.close()


def test_sampled_sa_walk_path(monkeypatch):
    """CS_FULL_SA=0: SAL walks bwt_invPsi from the 1-in-32 samples like the reference (bwt.c:86-96); same seeds"""
    import compseed_amd as ca
    monkeypatch.setenv("CS_FULL_SA", "0")
    monkeypatch.setenv("CS_MEM_CAP", "8")   # also force most reads through the overflow second pass
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0)
    for name, pname in (("repeat100", "default"), ("ragged", "k14"), ("main100", "c50s20")):
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
    assert e.stats()["overflow_reads"] > 0
    e.close(); ix.close()


@pytest.mark.parametrize("mode", ["fused", "split"])
def test_both_smem_kernels_match_golden(monkeypatch, mode):
    """the fused one-lane-per-read kernel and the forward / cooperative-backward kernels are two independent
    implementations of the same three rounds: both must reproduce the reference bit for bit"""
    import compseed_amd as ca
    monkeypatch.setenv("CS_SMEM_MODE", mode)
    monkeypatch.setenv("CS_MEM_CAP", "6")        # most reads overflow their first 6 slots: exercises both overflow paths
    monkeypatch.setenv("CS_LEP_ARENA_MB", "1")   # tiny LEP arena: the forward queue is processed in many chunks
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0)
    for name, pname in _data.golden_runs():
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
    e.close(); ix.close()


def test_text_mode_is_transparent(monkeypatch):
    """CS_TEXT_MODE=0 keeps unique matches on the FM index; the default finishes them on the 2-bit text and the inverse
    suffix array.  Same intervals, same seeds, same bwt_extend query count as the reference, fewer real index reads."""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    calls = {}
    monkeypatch.setenv("CS_R2_TEXT", "0")
    monkeypatch.setenv("CS_TEXT_SWEEP", "0")
    monkeypatch.setenv("CS_WINDOW", "0")
    for mode in ("0", "1"):
        monkeypatch.setenv("CS_TEXT_MODE", mode)
        e = ca.Engine(ix, 0)
        for name, pname in _data.golden_runs():
            z, kw = _data.load_golden(name, pname)
            bases, off = _data.load_reads(name)
            e.reset_stats()
            _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
            st = e.stats()
            assert st["bwt_queries"] == int(z["counters"][3]) or st["overflow_reads"] > 0
        e.reset_stats()
        bases, off = _data.load_reads("main100")
        e.seed_batch(bases, off, ca.Params())
        calls[mode] = e.stats()["bwt_calls"]
        e.close()
    ix.close()
    assert calls["1"] < calls["0"]


def test_reseeding_from_the_text_is_transparent(monkeypatch):
    """CS_R2_TEXT=0 runs every re-seeding call (bwamem.c:241-249) on the FM index; the default answers those of unique
    SMEMs from the repeat-length / LCP arrays when the text can decide them.  Same mems and seeds as the reference either
    way; with it on, some calls must actually have been answered from the text and fewer extensions evaluated."""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    tot = {}
    monkeypatch.setenv("CS_WINDOW", "0")
    monkeypatch.setenv("CS_TEXT_SWEEP", "0")
    for mode in ("0", "1"):
        monkeypatch.setenv("CS_R2_TEXT", mode)
        e = ca.Engine(ix, 0)
        e.reset_stats()
        for name, pname in _data.golden_runs():
            z, kw = _data.load_golden(name, pname)
            bases, off = _data.load_reads(name)
            _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
        tot[mode] = e.stats()
        e.close()
    ix.close()
    assert tot["0"]["reseed_text_calls"] == 0
    assert tot["1"]["reseed_text_calls"] > 0
    assert tot["1"]["bwt_calls"] < tot["0"]["bwt_calls"]
    assert tot["1"]["mems"] == tot["0"]["mems"] and tot["1"]["seeds"] == tot["0"]["seeds"]


def test_sweeps_read_off_the_text_are_transparent(monkeypatch):
    """CS_TEXT_SWEEP=0 runs every backward sweep (bwt.c:325-345) on the FM index; the default reads the sweep of a
    round-1 call off the text when its longest match is unique and agrees with the text back to the previous pivot."""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    tot = {}
    monkeypatch.setenv("CS_R2_TEXT", "0")
    monkeypatch.setenv("CS_WINDOW", "0")
    for mode in ("0", "1"):
        monkeypatch.setenv("CS_TEXT_SWEEP", mode)
        e = ca.Engine(ix, 0)
        e.reset_stats()
        for name, pname in _data.golden_runs():
            z, kw = _data.load_golden(name, pname)
            bases, off = _data.load_reads(name)
            _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
        tot[mode] = e.stats()
        e.close()
    ix.close()
    assert tot["0"]["sweep_text_calls"] == 0 and tot["1"]["sweep_text_calls"] > 0
    assert tot["1"]["bwt_queries"] < tot["0"]["bwt_queries"]
    assert tot["1"]["mems"] == tot["0"]["mems"] and tot["1"]["seeds"] == tot["0"]["seeds"]


def test_window_scheme_is_transparent(monkeypatch):
    """CS_WINDOW=0 sweeps every LEP backward in lockstep like bwt.c:325-345; the default settles the short ends through
    the k-mer jump table and lets every surviving end walk alone (smem_split.hpp, bwd_win_run).  Same mems, same seeds."""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    tot = {}
    monkeypatch.setenv("CS_R2_TEXT", "0")
    monkeypatch.setenv("CS_TEXT_SWEEP", "0")
    for mode in ("0", "1"):
        monkeypatch.setenv("CS_WINDOW", mode)
        e = ca.Engine(ix, 0)
        e.reset_stats()
        for name, pname in _data.golden_runs():
            z, kw = _data.load_golden(name, pname)
            bases, off = _data.load_reads(name)
            _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
        tot[mode] = e.stats()
        e.close()
    ix.close()
    assert tot["1"]["mems"] == tot["0"]["mems"] and tot["1"]["seeds"] == tot["0"]["seeds"]
    assert tot["1"]["bwt_calls"] < tot["0"]["bwt_calls"]


def test_round3_from_the_text_is_transparent(monkeypatch):
    """CS_R3_TEXT=0 computes every round-3 seed (bwt.c:357-381) on the FM index, beside rounds 1/2; the default runs
    round 3 afterwards and takes the seeds that lie inside a unique round-1 SMEM from the text arrays."""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    tot = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CS_R3_TEXT", mode)
        e = ca.Engine(ix, 0)
        e.reset_stats()
        for name, pname in _data.golden_runs():
            z, kw = _data.load_golden(name, pname)
            bases, off = _data.load_reads(name)
            _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
        tot[mode] = e.stats()
        e.close()
    ix.close()
    assert tot["0"]["r3_text_seeds"] == 0 and tot["1"]["r3_text_seeds"] > 0
    assert tot["1"]["mems"] == tot["0"]["mems"] and tot["1"]["seeds"] == tot["0"]["seeds"]
    assert tot["1"]["bwt_calls"] < tot["0"]["bwt_calls"]


def test_kmer_filter_is_transparent(monkeypatch):
    """CS_KMER_FILTER=0: the window lanes go straight to the jump table; default: they first ask a filter over all
    min_seed_len-mers of the text whether their window occurs at all ("no" is exact).  Same mems, fewer index reads."""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    tot = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CS_KMER_FILTER", mode)
        e = ca.Engine(ix, 0)
        e.reset_stats()
        for name, pname in _data.golden_runs():
            z, kw = _data.load_golden(name, pname)
            bases, off = _data.load_reads(name)
            _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
        tot[mode] = e.stats()
        e.close()
    ix.close()
    assert tot["1"]["mems"] == tot["0"]["mems"] and tot["1"]["seeds"] == tot["0"]["seeds"]
    assert tot["1"]["bwt_calls"] < tot["0"]["bwt_calls"]


@pytest.mark.parametrize("jk", ["0", "13", "9"])
def test_other_jump_table_sizes(monkeypatch, jk):
    """CS_JUMP_K: no jump table at all (forward passes and round 3 start base by base, no window scheme), or a smaller one
    (13-mers: six extensions from the table entry to min_seed_len; 9-mers: below the window scheme's range for -k 19)"""
    import compseed_amd as ca
    monkeypatch.setenv("CS_JUMP_K", jk)
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0)
    for name, pname in _data.golden_runs():
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
    e.close(); ix.close()


def test_cli_dump_matches_golden(tmp_path):
    """the CompSeed-compatible command line: same flags, seed dump identical to the reference golden"""
    import subprocess
    import compseed_amd as ca
    cli = os.path.join(os.path.dirname(ca.lib_path()), "compseed_amd_cli")
    out = tmp_path / "seeds.txt"
    r = subprocess.run([cli, "-t", "2", "-k", "14", "-K", "3000", "-w", "100", "-M", "--gpus", "1", "--dump-seeds", str(out), _data.PREFIX,
                        os.path.join(_data.GOLD, "ragged.txt")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    z, _ = _data.load_golden("ragged", "k14")
    mems, seeds = [], []
    for ln in open(out):
        f = ln.split("\t")
        if f[0] == "M":
            mems.append((int(f[1]) - 1, (int(f[2]) << 32) | int(f[3]), int(f[4]), int(f[5]), int(f[6])))
        else:
            seeds.append((int(f[1]) - 1, int(f[2]), int(f[3]), int(f[4])))
    gm = z["mems"]; rd = np.repeat(np.arange(z["mem_off"].size - 1), np.diff(z["mem_off"].astype(np.int64)))
    assert mems == [(int(rd[i]), int(gm[i, 3]), int(gm[i, 0]), int(gm[i, 1]), int(gm[i, 2])) for i in range(gm.shape[0])]
    rs = np.repeat(np.arange(z["seed_off"].size - 1), np.diff(z["seed_off"].astype(np.int64)))
    assert seeds == [(int(rs[i]), int(z["seed_qbeg"][i]), int(z["seed_len"][i]), int(z["seed_rbeg"][i])) for i in range(rs.size)]
    assert "BWT-extend:" in r.stderr and "SA Lookup:" in r.stderr


def test_device_sst_is_transparent(eng):
    """sst_mode switches the LDS-resident memo of bwt_extend on and off: identical seeds, fewer real calls when on
    (the cache only memoises a pure function -- SURVEY Appendix B.1 -- so the A/B is a built-in test)"""
    import compseed_amd as ca
    for name, pname in (("main100", "default"), ("ragged", "k14"), ("repeat100", "c50s20"), ("sorted150", "r1.0")):
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        stats = {}
        for mode in (0, 1):
            eng.reset_stats()
            res = eng.seed_batch(bases, off, ca.Params(sst_mode=mode, **kw))
            _check_against_golden(res, z)
            stats[mode] = eng.stats()
        assert stats[0]["bwt_queries"] == int(z["counters"][3]) and stats[0]["reseed_text_calls"] == 0
        assert stats[1]["bwt_calls"] < stats[0]["bwt_calls"]
        assert stats[0]["bwt_calls"] == stats[0]["bwt_queries"]
        assert stats[1]["bwt_calls"] < stats[1]["bwt_queries"]
