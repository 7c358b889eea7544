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
    if st["overflow_mems"] == 0:                        # (reads that overflow into the fused second pass are seeded twice)
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


NT4_LUT = np.full(256, 4, np.uint8)   # nst_nt4_table restricted to what pack_reads_kernel distinguishes: A C G T in either case
for _i, _ch in enumerate(b"ACGT"):
    NT4_LUT[_ch] = _i; NT4_LUT[_ch | 0x20] = _i


def test_unaligned_and_coded_device_inputs(eng):
    """pack_reads_kernel reads the caller's bytes in aligned 8-byte words when the pointer allows it and goes through the nt4 copy
    when it does not; ASCII (either case, IUPAC, '-') and 0..4 codes are the same reads (comp_seed.cpp:2258-2260)"""
    import compseed_amd as ca
    for name in ("ragged", "main100"):
        bases, off = _data.load_reads(name)
        want = eng.seed_batch(bases, off)
        d_o = eng.alloc(off.nbytes); eng.upload(d_o, off)
        for shift in (0, 3):
            for coded in (False, True):
                b = NT4_LUT[bases] if coded else bases
                d_b = eng.alloc(b.nbytes + 80)
                eng.upload(d_b + shift, b)
                dev = eng.seed_batch_device(d_b + shift, d_o, off.size - 1, b.size)
                assert dev.n_mems == want.n_mems and dev.n_seeds == want.n_seeds, (name, shift, coded)
                assert np.array_equal(eng.download(dev.ptr["mem_off"], np.uint64, off.size), want.mem_off)
                assert np.array_equal(eng.download(dev.ptr["mems"], ca.INTV_DT, dev.n_mems), want.mems), (name, shift, coded)
                assert np.array_equal(eng.download(dev.ptr["seeds"], ca.SEED_DT, dev.n_seeds), want.seeds), (name, shift, coded)
                assert np.array_equal(eng.download(d_b + shift, np.uint8, b.size), b)      # untouched
                eng.free(d_b)
        eng.free(d_o)


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


def test_sampled_sa_walk_path():
    """full_sa=0: SAL walks bwt_invPsi from the 1-in-32 samples like the reference (bwt.c:86-96); same seeds"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0, full_sa=0, mem_cap=8)   # mem_cap 8: also force most reads through the overflow records
    for name, pname in (("repeat100", "default"), ("ragged", "k14"), ("main100", "c50s20")):
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
    assert e.stats()["overflow_mems"] > 0
    e.close(); ix.close()


@pytest.mark.parametrize("fused", [1, 0])
def test_both_smem_kernels_match_golden(fused):
    """the fused one-lane-per-read kernel and the forward / cooperative-backward kernels are two independent
    implementations of the same three rounds: both must reproduce the reference bit for bit"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    # mem_cap 6: most reads overflow their first 6 slots (both overflow paths); 1 MB LEP arena: the forward queue is processed in many chunks
    e = ca.Engine(ix, 0, fused=fused, mem_cap=6, lep_arena_mb=1)
    for name, pname in _data.golden_runs():
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
    e.close(); ix.close()


@pytest.fixture(scope="module")
def eng64():
    """the instantiation hg19 scale runs: 8-byte suffix-array and inverse-suffix-array entries (rows >= 2^32 there)"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0, sa64=1)
    yield e
    e.close()
    ix.close()


@pytest.mark.parametrize("name,pname", _data.golden_runs())
def test_golden_seeds_64bit_text_side(eng64, name, pname):
    """every reference golden through fsa64 / isa64 / lcp / rep as the benchmark's index uses them"""
    import compseed_amd as ca
    z, kw = _data.load_golden(name, pname)
    bases, off = _data.load_reads(name)
    eng64.reset_stats()
    _check_against_golden(eng64.seed_batch(bases, off, ca.Params(**kw)), z)
    off_all = eng64.seed_batch(bases, off, ca.Params(sst_mode=0, **kw))
    _check_against_golden(off_all, z)


def test_64bit_text_side_mechanisms_fire_and_sa_agrees(eng64):
    import compseed_amd as ca
    eng64.reset_stats()
    for name in ("main100", "sorted150"):
        bases, off = _data.load_reads(name)
        eng64.seed_batch(bases, off, ca.Params())
    st = eng64.stats()
    assert st["reseed_text_calls"] > 0 and st["sweep_text_calls"] > 0 and st["r3_text_seeds"] > 0
    n = int(eng64._index.view.seq_len)
    got = eng64.sa(np.arange(0, n + 1, dtype=np.uint64))          # sa_kernel poisons rows where fsa64 != the walk
    assert not (got == np.uint64(0xdeadbeefdeadbeef)).any()


def _run_goldens(e, **pkw):
    import compseed_amd as ca
    e.reset_stats()
    per_run = {}
    for name, pname in _data.golden_runs():
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        q0 = e.stats()["bwt_queries"]
        _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw, **pkw)), z)
        per_run[(name, pname)] = (e.stats()["bwt_queries"] - q0, int(z["counters"][3]))
    return e.stats(), per_run


# every exact shortcut of DESIGN.md section 4.2: (switch, the others that must be off so that the effect is isolated, what must
# have fired with it on, which counter must drop)
MECHANISMS = {
    "text_mode": (("r2_text", "text_sweep", "window", "r3_text"), None, "bwt_calls"),
    "r2_text": (("window", "text_sweep"), "reseed_text_calls", "bwt_calls"),
    "text_sweep": (("r2_text", "window"), "sweep_text_calls", "bwt_queries"),
    "window": (("r2_text", "text_sweep"), None, "bwt_calls"),
    "r3_text": ((), "r3_text_seeds", "bwt_calls"),
    "kmer_filter": ((), None, "bwt_calls"),
    "fwd0": ((), None, None),
}


@pytest.mark.parametrize("mech", sorted(MECHANISMS))
def test_each_shortcut_is_transparent(eng, mech):
    """cs_params_t.disable switches one mechanism off: all 14 reference goldens with it off and on, identical mems and seeds,
    and the mechanism must actually have fired / saved index reads"""
    import compseed_amd as ca
    others, fired, drops = MECHANISMS[mech]
    base = ca.binding.disable_mask(*others)
    st_off, runs_off = _run_goldens(eng, disable=base | ca.binding.disable_mask(mech))
    st_on, _ = _run_goldens(eng, disable=base)
    assert st_on["mems"] == st_off["mems"] and st_on["seeds"] == st_off["seeds"]
    if fired:
        assert st_off[fired] == 0 and st_on[fired] > 0
    if drops:
        assert st_on[drops] < st_off[drops]
    if mech == "text_mode":  # with every query-skipping shortcut off the query count is the reference's, run by run
        for key, (got, want) in runs_off.items():
            assert got == want, key


def test_all_shortcuts_off_by_flags_equals_reference_count(eng):
    import compseed_amd as ca
    st, runs = _run_goldens(eng, disable=0xffffffff)
    assert st["reseed_text_calls"] == 0 and st["sweep_text_calls"] == 0 and st["r3_text_seeds"] == 0


@pytest.mark.parametrize("jk", [0, 13, 9])
def test_other_jump_table_sizes(jk):
    """jump_k: no jump table at all (forward passes and round 3 start base by base, no window scheme), or a smaller one
    (13-mers: six extensions from the table entry to min_seed_len; 9-mers: below the window scheme's range for -k 19)"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0, jump_k=jk)
    _run_goldens(e)
    e.close(); ix.close()


def test_without_optional_arrays():
    """the degrade path of a part with little HBM: no text arrays, no text, no full SA -- each level must still be exact"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    for opts in (dict(text_arrays=0), dict(text_mode=0), dict(full_sa=0), dict(full_sa=0, jump_k=0, kmer_filter=0)):
        e = ca.Engine(ix, 0, **opts)
        st, _ = _run_goldens(e)
        if "text_arrays" in opts or "text_mode" in opts or "full_sa" in opts:
            assert st["reseed_text_calls"] == 0 and st["r3_text_seeds"] == 0
        e.close()
    ix.close()


def test_sal_merged_count_is_the_references():
    """count_sal_merged: cs_stats_t.sal_calls = distinct SA slots per 512 reads, CompSeed's 'SA Lookup ... calls' (comp_seed.cpp:2327-2345)"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0, count_sal_merged=1)
    for name, pname in _data.golden_runs():
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        e.reset_stats()
        _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
        st = e.stats()
        assert st["sal_queries"] == int(z["counters"][5]) and st["sal_calls"] == int(z["counters"][6]), (name, pname)
    e.close(); ix.close()


def test_device_offsets_are_validated(eng):
    """cs_engine_seed_batch_device: offsets that do not tile [0, n_bases) are CS_EINVAL, not an out-of-bounds read"""
    import compseed_amd as ca
    bases, off = _data.load_reads("sorted150")
    d_b = eng.alloc(bases.nbytes + 64); eng.upload(d_b, bases)
    n = off.size - 1
    for bad in (off + np.uint64(1), np.concatenate([off[:-1], [off[-1] + np.uint64(4096)]]).astype(np.uint64),
                np.concatenate([off[:5], [off[3]], off[6:]]).astype(np.uint64)):
        d_o = eng.alloc(bad.nbytes); eng.upload(d_o, bad)
        with pytest.raises(ca.CSError) as ei:
            eng.seed_batch_device(d_b, d_o, n, bases.size)
        assert ei.value.code == -1
        eng.free(d_o)
    d_o = eng.alloc(off.nbytes); eng.upload(d_o, off)
    assert eng.seed_batch_device(d_b, d_o, n, bases.size).n_mems > 0      # and the engine is still usable afterwards
    eng.free(d_o); eng.free(d_b)


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


def test_cli_shards_over_two_gpus_when_present(tmp_path):
    """compseed_amd_cli --gpus 2: every chunk split into two contiguous read ranges, one engine per GPU; same dump as with one GPU
    (skipped on a one-GPU box; torch.cuda.device_count() does not initialise the runtime here)"""
    import subprocess
    import torch
    import compseed_amd as ca
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    cli = os.path.join(os.path.dirname(ca.lib_path()), "compseed_amd_cli")
    outs = []
    for g in ("1", "2"):
        out = tmp_path / ("seeds%s.txt" % g)
        r = subprocess.run([cli, "-K", "30000", "--gpus", g, "--dump-seeds", str(out), _data.PREFIX, os.path.join(_data.GOLD, "main100.txt")],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].count("\n") > 10000


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


def test_result_digest_and_gather_reads(eng):
    """cs_engine_result_digest == the host restatement over the downloaded arrays; cs_engine_gather_reads == slices of the full result"""
    import compseed_amd as ca
    bases, off = _data.load_reads("main100")
    full = eng.seed_batch(bases, off)
    want = tuple(ca.binding.digest_words(a) for a in (full.mem_off, full.mems, full.seed_off, full.seeds))
    assert eng.result_digest() == want
    ids = np.array([2999, 0, 7, 7, 1500, 2998, 1], dtype=np.uint64)
    g = eng.gather_reads(ids)
    assert g.n_reads == ids.size
    for j, r in enumerate(ids.astype(int)):
        assert np.array_equal(g.mems[int(g.mem_off[j]):int(g.mem_off[j + 1])], full.mems[int(full.mem_off[r]):int(full.mem_off[r + 1])])
        assert np.array_equal(g.seeds[int(g.seed_off[j]):int(g.seed_off[j + 1])], full.seeds[int(full.seed_off[r]):int(full.seed_off[r + 1])])
    nos = eng.seed_batch(bases, off, ca.Params(want_sal=0))
    d = eng.result_digest()
    assert d[0] == want[0] and d[1] == want[1] and d[2] == 0 and d[3] == 0
    assert eng.gather_reads(ids).seeds is None and np.array_equal(eng.gather_reads(ids).mems, g.mems) and nos.n_seeds == 0
    with pytest.raises(ca.CSError):
        eng.gather_reads(np.array([3000], dtype=np.uint64))


def test_config1_ecoli_size_set_on_the_gpu():
    """BASELINE.json configs[0] on the device: index built on the GPU == bwaidx's (by md5), seeds == the reference's (by digest),
    and with every shortcut off the device evaluates exactly the reference's 38,109,585 bwt_extend queries"""
    import hashlib
    import json
    import compseed_amd as ca
    want = json.load(open(os.path.join(_data.HERE, "golden", "c1", "config1.json")))
    ref, reads, shuf = _data.config1_dataset()
    ix = ca.Index.build(_data.codes_of(ref), 0)
    bw, sa = ix.arrays()
    v = ix.view
    hdr = np.array([v.primary] + [v.L2[i] for i in range(1, 5)], dtype="<u8").tobytes()
    assert hashlib.md5(hdr + bw.tobytes()).hexdigest() == want["index_md5"]["bwt"]
    e = ca.Engine(ix, 0, count_sal_merged=1)
    for order, rd in (("sorted", reads), ("shuffled", shuf)):
        w = want["sets"][order]
        bases, off = _data.pack_reads([r.encode() for r in rd])
        for sst in (1, 0):
            e.reset_stats()
            got = e.seed_batch(bases, off, ca.Params(sst_mode=sst))
            m = got.mems
            d = _data.digest_result(got.mem_off, np.stack([m["x0"], m["x1"], m["x2"], m["info"]], axis=1), got.seed_off,
                                    got.seeds["rbeg"], got.seeds["qbeg"], got.seeds["len"])
            assert d == w["digest"], (order, sst)
            st = e.stats()
            assert st["sal_queries"] == w["sal_queries"] and st["sal_calls"] == w["sal_calls"]
            if sst == 0:
                assert st["bwt_queries"] == w["bwt_queries"] == 38109585
    e.close(); ix.close()


def _expand_packed(p):
    """host restatement of what cs_engine_seed_batch does with the packed form (and what a consumer does with cs_unpack_mem)"""
    import compseed_amd as ca
    mems = ca.unpack_mems16(p["mems"]) if p["mem_format"] == 1 else p["mems"].copy()
    if p["seed_off"] is None:
        return mems, None
    cnt = np.minimum(mems["x2"], np.uint64(p["max_occ"])).astype(np.int64)
    seeds = np.zeros(p["n_seeds"], dtype=ca.SEED_DT)
    seeds["rbeg"] = p["seed_rbeg"]
    seeds["qbeg"] = np.repeat((mems["info"] >> np.uint64(32)).astype(np.int32), cnt)
    seeds["len"] = np.repeat(((mems["info"] & np.uint64(0xffffffff)) - (mems["info"] >> np.uint64(32))).astype(np.int32), cnt)
    return mems, seeds


@pytest.mark.parametrize("pipeline_reads,host_pack_threads", [(0, 8), (700, 3), (64, 1), (700, 0)])
def test_packed_and_pipelined_host_variants(pipeline_reads, host_pack_threads):
    """cs_engine_seed_batch_packed (16-byte mems, seeds as rbeg) and cs_engine_seed_batch (the same, expanded on host threads), cut into
    sub-batches of `pipeline_reads` reads whose upload / seeding / download overlap: every golden run, bit for bit, from pageable and
    from pinned input (cs_host_alloc)"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0, pipeline_reads=pipeline_reads, expand_threads=3, host_pack_threads=host_pack_threads)   # (0: the caller's bytes go up and the GPU makes the records)
    for name, pname in _data.golden_runs():
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        _check_against_golden(e.seed_batch(bases, off, ca.Params(**kw)), z)
        pin = ca.pinned_array(max(1, bases.size))
        pin[:bases.size] = bases
        p = e.seed_batch_packed(pin, off, ca.Params(**kw))
        assert p["mem_format"] == 1 and p["mems"].dtype.itemsize == 16
        mems, seeds = _expand_packed(p)
        assert np.array_equal(p["mem_off"], z["mem_off"]) and np.array_equal(p["seed_off"], z["seed_off"])
        assert np.array_equal(np.stack([mems["x0"], mems["x1"], mems["x2"], mems["info"]], axis=1), z["mems"])
        assert np.array_equal(seeds["rbeg"], z["seed_rbeg"]) and np.array_equal(seeds["qbeg"], z["seed_qbeg"]) and np.array_equal(seeds["len"], z["seed_len"])
        nos = e.seed_batch_packed(bases, off, ca.Params(want_sal=0, **kw))
        assert nos["seed_off"] is None and nos["n_seeds"] == 0 and np.array_equal(ca.unpack_mems16(nos["mems"]), mems)
        r0 = e.seed_batch(bases, off, ca.Params(want_sal=0, **kw))
        assert r0.seeds is None and np.array_equal(r0.mems, mems)
    e.close(); ix.close()


def test_host_packed_reads_through_the_fused_kernel():
    """a split width beyond the task field sends a batch to the fused kernel, which reads a byte per base: made back from the records
    the host uploaded (unpack_reads_kernel); same results as with the caller's bytes uploaded, and as the oracle's"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    bases, off = _data.load_reads("ragged")
    par = ca.Params(s=20000)
    res = []
    for hpt in (4, 0):
        e = ca.Engine(ix, 0, host_pack_threads=hpt)
        res.append(e.seed_batch(bases, off, par))
        e.close()
    assert np.array_equal(res[0].mems, res[1].mems) and np.array_equal(res[0].seeds, res[1].seeds) and np.array_equal(res[0].mem_off, res[1].mem_off)
    o = _oracle.OracleIndex(_data.PREFIX)
    want = o.seed_batch(bases, off, _oracle.make_params(s=20000), mode=0, threads=2)
    assert np.array_equal(res[0].mems, want["mems"]) and np.array_equal(res[0].seeds, want["seeds"])
    o.close(); ix.close()


def test_packed_falls_back_to_full_records_for_long_reads(eng):
    """reads of 2^15 bases or more do not fit the 15-bit query fields of cs_mem16_t: the packed call returns plain cs_intv_t records"""
    import compseed_amd as ca
    import gzip
    fa = gzip.open(os.path.join(_data.GOLD, "ref.fa.gz")).read().decode().split("\n")
    g = "".join(l for l in fa if not l.startswith(">")).replace("N", "A")
    bases, off = _data.pack_reads([g[1000:41000].encode(), g[50000:50150].encode()])
    p = eng.seed_batch_packed(bases, off)
    assert p["mem_format"] == 0 and p["mems"].dtype == ca.INTV_DT
    full = eng.seed_batch(bases, off)
    assert np.array_equal(p["mems"], full.mems) and np.array_equal(p["seed_rbeg"], full.seeds["rbeg"]) and np.array_equal(p["mem_off"], full.mem_off)
    o = _oracle.OracleIndex(_data.PREFIX)
    want = o.seed_batch(bases, off, mode=0, threads=2)
    assert np.array_equal(full.mems, want["mems"]) and np.array_equal(full.seeds, want["seeds"])
    o.close()


def test_submit_collect_pipeline_across_batches():
    """cs_engine_submit / cs_engine_collect_packed: batches in flight (four at most), results come back in submission order and equal the goldens;
    a collected result stays intact while later batches are submitted, seeded and downloaded; misuse is an error code"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0, pipeline_reads=900)
    runs = [("main100", "default"), ("sorted150", "default"), ("repeat100", "default"), ("ragged", "default"), ("main100", "default"), ("shuffled100", "default")]
    data = []
    for name, pname in runs:
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        pin = ca.pinned_array(max(1, bases.size)); pin[:bases.size] = bases
        data.append((z, pin, off))
    with pytest.raises(ca.CSError):
        e.collect_packed()                                   # nothing submitted
    e.submit(data[0][1], data[0][2]); e.submit(data[1][1], data[1][2]); e.submit(data[2][1], data[2][2]); e.submit(data[3][1], data[3][2])
    with pytest.raises(ca.CSError):
        e.submit(data[4][1], data[4][2])                     # four in flight already
    with pytest.raises(ca.CSError):
        e.seed_batch(data[2][1], data[2][2])                 # blocking calls refuse while batches are in flight
    held = None
    for i in range(len(runs)):
        p = e.collect_packed()
        if held is not None:                                 # (the previous result was valid until this collect: checked below before it)
            pass
        z = data[i][0]
        mems, seeds = _expand_packed(p)
        assert np.array_equal(p["mem_off"], z["mem_off"]) and np.array_equal(np.stack([mems["x0"], mems["x1"], mems["x2"], mems["info"]], axis=1), z["mems"]), runs[i]
        assert np.array_equal(seeds["rbeg"], z["seed_rbeg"]) and np.array_equal(p["seed_off"], z["seed_off"]), runs[i]
        if i + 4 < len(runs):
            e.submit(data[i + 4][1], data[i + 4][2])         # keep four in flight
            import time
            time.sleep(0.05)                                 # let the next batches run: the held result must not change under us
            m2, s2 = _expand_packed(p)
            assert np.array_equal(m2, mems) and np.array_equal(s2["rbeg"], seeds["rbeg"])
        held = p
    got = e.seed_batch(data[0][1], data[0][2])               # and the blocking calls work again afterwards
    _check_against_golden(got, data[0][0])
    e.close(); ix.close()


@pytest.mark.parametrize("name,pname", [("main100", "default"), ("repeat100", "default"), ("ragged", "k14"), ("sorted150", "r1.0"), ("main100", "c50s20")])
def test_engine_output_chains_like_the_reference(eng, name, pname):
    """seed -> chain hand-off on the engine's own output: cs_engine_seed_batch -> cs_chain_batch == the reference's mem_chain (golden chains)"""
    import compseed_amd as ca
    from test_chain import check_chains, golden_chains
    z, kw = _data.load_golden(name, pname)
    bases, off = _data.load_reads(name)
    r = eng.seed_batch(bases, off, ca.Params(**kw))
    c = ca.Chainer(_data.PREFIX)
    check_chains(c.chain(r.mem_off, r.mems, r.seed_off, r.seeds, off, ca.ChainParams(k=kw.get("k", 19), c=kw.get("c", 500)), threads=4), golden_chains(name, pname))
    c.close()


def test_cli_shards_over_two_engines_on_one_gpu(tmp_path):
    """the sharded path of the C++ drop-in on a one-GPU box: --devices 0,0 puts both shards' engines on the same GPU (every chunk split into
    two contiguous read ranges, two engines, two submit / collect pipelines at once); same dump as one engine, also with -K chunks"""
    import subprocess
    import compseed_amd as ca
    cli = os.path.join(os.path.dirname(ca.lib_path()), "compseed_amd_cli")
    outs = []
    for dev in (("--gpus", "1"), ("--devices", "0,0"), ("--devices", "0,0,0")):
        out = tmp_path / ("seeds%d.txt" % len(outs))
        r = subprocess.run([cli, "-K", "30000", *dev, "--dump-seeds", str(out), _data.PREFIX, os.path.join(_data.GOLD, "main100.txt")],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert ("on %d GPU(s)" % (1 if dev[0] == "--gpus" else dev[1].count(",") + 1)) in r.stderr
        outs.append(open(out).read())
    assert outs[0] == outs[1] == outs[2] and outs[0].count("\n") > 10000


def test_submit_growing_batches_while_the_other_is_in_flight():
    """a batch that needs LARGER input / offset buffers is submitted while the previous one is still somewhere in the
    seeding thread (seeded, its pack kernels and downloads not yet queued).  The reallocation has to wait for it: every result must still be
    the golden's.  Batches grow from 40 reads to 3000 and shrink again, two in flight throughout, several rounds, on a fresh engine each
    round (a DevBuf has no slack on its first allocation, which is when the race was open)."""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    z, _ = _data.load_golden("main100", "default")
    bases, off = _data.load_reads("main100")
    n = off.size - 1
    want_m = z["mems"]; mo = z["mem_off"].astype(np.int64); so = z["seed_off"].astype(np.int64)

    def sub(k0, k1):
        pin = ca.pinned_array(int(off[k1] - off[k0])); pin[:] = bases[int(off[k0]):int(off[k1])]
        return pin, (off[k0:k1 + 1] - off[k0]).astype(np.uint64), (k0, k1)
    sizes = [40, 90, 400, 1300, 3000, 700, 2900, 60, 3000]
    for rnd in range(3):
        e = ca.Engine(ix, 0, pipeline_reads=350 if rnd else 5000000)
        jobs = []
        for j, sz in enumerate(sizes):
            k0 = (j * 131 + rnd * 17) % (n - sz + 1)
            jobs.append(sub(k0, k0 + sz))
        depth = 2 + rnd                                                            # two in flight (what the reference's kt_pipeline keeps), three, four
        for j in range(depth):
            e.submit(jobs[j][0], jobs[j][1])
        for i in range(len(jobs)):
            p = e.collect_packed()
            k0, k1 = jobs[i][2]
            mems, seeds = _expand_packed(p)
            assert np.array_equal(p["mem_off"].astype(np.int64), mo[k0:k1 + 1] - mo[k0]), (rnd, i)
            assert np.array_equal(np.stack([mems["x0"], mems["x1"], mems["x2"], mems["info"]], axis=1), want_m[mo[k0]:mo[k1]]), (rnd, i)
            assert np.array_equal(seeds["rbeg"], z["seed_rbeg"][so[k0]:so[k1]]), (rnd, i)
            if i + depth < len(jobs):
                e.submit(jobs[i + depth][0], jobs[i + depth][1])
        e.close()
    ix.close()


def _download_result(eng, r, n_reads):
    import compseed_amd as ca
    mo = eng.download(r.ptr["mem_off"], np.uint64, n_reads + 1)
    mm = eng.download(r.ptr["mems"], ca.INTV_DT, r.n_mems)
    so = eng.download(r.ptr["seed_off"], np.uint64, n_reads + 1)
    ss = eng.download(r.ptr["seeds"], ca.SEED_DT, r.n_seeds)
    return mo, mm, so, ss


@pytest.mark.parametrize("passes", [2, 1])
def test_device_batches_in_flight_on_two_pass_contexts(passes):
    """cs_engine_submit_device / cs_engine_collect_device: every golden run as a stream of device-resident batches, two in flight on
    alternating pass contexts (own -k / -r / -y / -c / -s per batch: the k-mer filter is rebuilt between passes that run at the same
    time); results in submission order, bit for bit; a collected result stays valid while the next batch runs; misuse is an error code"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    e = ca.Engine(ix, 0, passes_in_flight=passes)
    runs = _data.golden_runs()
    dev = []
    for name, pname in runs:
        z, kw = _data.load_golden(name, pname)
        bases, off = _data.load_reads(name)
        d_b = e.alloc(bases.nbytes + 64); d_o = e.alloc(off.nbytes)
        e.upload(d_b, bases); e.upload(d_o, off)
        dev.append((z, kw, d_b, d_o, off.size - 1, bases.size))
    with pytest.raises(ca.CSError):
        e.collect_device()                                   # nothing submitted

    def sub(i):
        z, kw, d_b, d_o, n, nb = dev[i]
        e.submit_device(d_b, d_o, n, nb, ca.Params(**kw))
    for i in range(passes):
        sub(i)
    with pytest.raises(ca.CSError):
        sub(passes)                                          # that many are in flight already
    with pytest.raises(ca.CSError):
        e.seed_batch_device(dev[0][2], dev[0][3], dev[0][4], dev[0][5])      # blocking calls refuse meanwhile
    host_b, host_o = _data.load_reads(runs[0][0])
    with pytest.raises(ca.CSError):
        e.submit(host_b, host_o)                             # and so does the host pipeline
    for i in range(len(runs)):
        r = e.collect_device()
        if i + passes < len(runs):
            sub(i + passes)                                  # the next batch starts; r's context is not reused before the submit after this one
        z, n = dev[i][0], dev[i][4]
        mo, mm, so, ss = _download_result(e, r, n)
        assert np.array_equal(mo, z["mem_off"]) and np.array_equal(np.stack([mm["x0"], mm["x1"], mm["x2"], mm["info"]], axis=1), z["mems"]), runs[i]
        assert np.array_equal(so, z["seed_off"]) and np.array_equal(ss["rbeg"], z["seed_rbeg"]) and np.array_equal(ss["qbeg"], z["seed_qbeg"]) and np.array_equal(ss["len"], z["seed_len"]), runs[i]
    dg = e.result_digest()                                   # the last batch is still held by its context
    z, kw, d_b, d_o, n, nb = dev[-1]
    e.seed_batch_device(d_b, d_o, n, nb, ca.Params(**kw))
    assert e.result_digest() == dg
    st = e.stats()
    assert st["reads"] == sum(d[4] for d in dev) + n
    for d in dev:
        e.free(d[2]); e.free(d[3])
    e.close(); ix.close()


def test_stream_of_host_batches_with_changing_parameters():
    """cs_engine_submit / collect: all golden runs through the host pipeline, three in flight, whole and cut into parts that are packed in
    order; consecutive batches differ in -k, so the k-mer filter is rebuilt between passes (under the lock the pass contexts share)"""
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    for pr in (700, 0):
        e = ca.Engine(ix, 0, pipeline_reads=pr)
        runs = _data.golden_runs()
        data = []
        for name, pname in runs:
            z, kw = _data.load_golden(name, pname)
            bases, off = _data.load_reads(name)
            data.append((z, kw, bases, off))
        for i in range(3):
            e.submit(data[i][2], data[i][3], ca.Params(**data[i][1]))
        for i in range(len(runs)):
            p = e.collect_packed()
            z = data[i][0]
            mems, seeds = _expand_packed(p)
            assert np.array_equal(p["mem_off"], z["mem_off"]) and np.array_equal(np.stack([mems["x0"], mems["x1"], mems["x2"], mems["info"]], axis=1), z["mems"]), runs[i]
            assert np.array_equal(seeds["rbeg"], z["seed_rbeg"]) and np.array_equal(p["seed_off"], z["seed_off"]), runs[i]
            if i + 3 < len(runs):
                e.submit(data[i + 3][2], data[i + 3][3], ca.Params(**data[i + 3][1]))
        e.close()
    ix.close()
