"""Size-independent properties of the hot path at sizes the golden vectors cannot cover (GPU, through the C ABI).

A 64 Mbp synthetic genome is indexed on the GPU and 400,000 x 150 bp reads are seeded; the checks need no oracle
run over the full set: sortedness and filters, every seed position verified against the text, strand symmetry of the
bi-intervals, invariance under batch splitting / SST mode / kernel variant, and a bit-exact oracle comparison on a sample.
The last test repeats the cheap properties at BASELINE.json's full size (hg19-scale index, 10 M x 150 bp reads).
"""
import os
import sys

import numpy as np
import pytest

import _oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def world():
    import torch
    import compseed_amd as ca
    import synth
    G = synth.make_genome(64_000_000, seed=11, device="cuda")
    g = G.cpu().numpy()
    ix = ca.Index.build(g, 0)
    eng = ca.Engine(ix, 0)
    bases, off = synth.make_reads(G, 400_000, 150, seed=5, p_sub=0.01, p_n=0.0005, sort=True)
    hb, ho = bases.cpu().numpy(), off.cpu().numpy().astype(np.uint64)
    del G
    torch.cuda.empty_cache()
    res = eng.seed_batch(hb, ho)
    yield dict(ca=ca, g=g, ix=ix, eng=eng, bases=hb, off=ho, res=res)
    eng.close(); ix.close()


def _text(g):
    return np.concatenate([g, (3 - g[::-1]).astype(np.uint8)])  # forward ++ reverse complement (bntseq.c:306-312)


def _codes(ascii_bases):
    lut = np.full(256, 4, np.uint8)
    for ch, v in zip(b"ACGT", range(4)):
        lut[ch] = v
    return lut[ascii_bases]


def test_sorted_filtered_and_consistent(world):
    r = world["res"]
    m = r.mems
    beg = (m["info"] >> np.uint64(32)).astype(np.int64); end = (m["info"] & np.uint64(0xffffffff)).astype(np.int64)
    assert (end - beg >= 19).all() and (m["x2"] >= 1).all()                       # length filter, non-empty intervals
    assert (end <= 150).all() and (beg >= 0).all()
    rd = np.repeat(np.arange(r.n_reads), np.diff(r.mem_off.astype(np.int64)))
    same = rd[1:] == rd[:-1]
    assert (m["info"][1:][same] >= m["info"][:-1][same]).all()                    # sorted by info inside a read (comp_seed.cpp:2301)
    n = world["ix"].view.seq_len
    assert (m["x0"] + m["x2"] <= n + 1).all() and (m["x1"] + m["x2"] <= n + 1).all()
    # SAL: min(x2, max_occ) slots per mem, in mem order (comp_seed.cpp:2313-2325)
    per_mem = np.minimum(m["x2"], 500).astype(np.int64)
    assert per_mem.sum() == r.n_seeds
    so = np.concatenate([[0], np.cumsum(per_mem)])
    assert np.array_equal(r.seed_off.astype(np.int64), so[r.mem_off.astype(np.int64)])
    assert np.array_equal(r.seeds["qbeg"], np.repeat(beg, per_mem)) and np.array_equal(r.seeds["len"], np.repeat(end - beg, per_mem))


def test_every_seed_matches_the_text(world):
    """end-to-end: read[qbeg:qbeg+len] == T[rbeg:rbeg+len] for every one of the ~4 M seeds (checks x0 and the SA path)"""
    r = world["res"]; T = _text(world["g"]); q = _codes(world["bases"])
    rd = np.repeat(np.arange(r.n_reads), np.diff(r.seed_off.astype(np.int64)))
    qpos = world["off"][rd].astype(np.int64) + r.seeds["qbeg"]
    rbeg = r.seeds["rbeg"].astype(np.int64); ln = r.seeds["len"].astype(np.int64)
    assert (rbeg >= 0).all() and (rbeg + ln <= T.size).all()
    ok = np.ones(rbeg.size, bool)
    for d in range(int(ln.max())):
        act = ln > d
        ok[act] &= T[rbeg[act] + d] == q[qpos[act] + d]
    assert ok.all()


def test_strand_symmetry(world):
    """the SMEM set of a read (round 1 only: -y 0 disables round 3, a huge -r disables re-seeding, both of which walk the
    read left to right and are not symmetric) is mirrored by reverse-complementing the read: x0 <-> x1, [beg,end) -> [L-end, L-beg)"""
    ca, eng = world["ca"], world["eng"]
    n = 20000
    q = world["bases"][: n * 150].reshape(n, 150)
    comp = np.zeros(256, np.uint8); comp[list(b"ACGTN")] = list(b"TGCAN")
    rc = comp[q[:, ::-1]].reshape(-1)
    off = world["off"][: n + 1]
    a = eng.seed_batch(q.reshape(-1), off, ca.Params(want_sal=0, y=0, r=100.0))
    b = eng.seed_batch(rc, off, ca.Params(want_sal=0, y=0, r=100.0))
    assert np.array_equal(a.mem_off, b.mem_off)
    ka = np.stack([np.repeat(np.arange(n), np.diff(a.mem_off.astype(np.int64))), 150 - (a.mems["info"] & np.uint64(0xffffffff)).astype(np.int64),
                   150 - (a.mems["info"] >> np.uint64(32)).astype(np.int64), a.mems["x1"].astype(np.int64), a.mems["x0"].astype(np.int64), a.mems["x2"].astype(np.int64)], 1)
    kb = np.stack([np.repeat(np.arange(n), np.diff(b.mem_off.astype(np.int64))), (b.mems["info"] >> np.uint64(32)).astype(np.int64),
                   (b.mems["info"] & np.uint64(0xffffffff)).astype(np.int64), b.mems["x0"].astype(np.int64), b.mems["x1"].astype(np.int64), b.mems["x2"].astype(np.int64)], 1)
    ka = ka[np.lexsort(ka.T[::-1])]; kb = kb[np.lexsort(kb.T[::-1])]
    assert np.array_equal(ka, kb)


def test_invariance_under_batching_sst_and_kernel_variant(world):
    ca, eng, res = world["ca"], world["eng"], world["res"]
    hb, ho = world["bases"], world["off"]
    again = eng.seed_batch(hb, ho)                                          # determinism despite atomics / task order
    assert np.array_equal(again.mems, res.mems) and np.array_equal(again.seeds, res.seeds)
    off_sst = eng.seed_batch(hb, ho, ca.Params(sst_mode=0))
    assert np.array_equal(off_sst.mems, res.mems) and np.array_equal(off_sst.mem_off, res.mem_off)
    cut = 123_457                                                           # results of a read do not depend on its batch
    p1 = eng.seed_batch(hb[: cut * 150], ho[: cut + 1])
    p2 = eng.seed_batch(hb[cut * 150:], ho[cut:] - ho[cut])
    assert np.array_equal(np.concatenate([p1.mems, p2.mems]), res.mems) and np.array_equal(np.concatenate([p1.seeds, p2.seeds]), res.seeds)
    e3 = ca.Engine(world["ix"], 0, pipeline_reads=90_000, expand_threads=4)  # the host pipeline: 4 sub-batches, unequal tails
    pp = e3.seed_batch(hb, ho)
    assert np.array_equal(pp.mem_off, res.mem_off) and np.array_equal(pp.mems, res.mems) and np.array_equal(pp.seed_off, res.seed_off) and np.array_equal(pp.seeds, res.seeds)
    pk = e3.seed_batch_packed(hb, ho)
    assert pk["n_mems"] == res.n_mems and np.array_equal(ca.unpack_mems16(pk["mems"]), res.mems) and np.array_equal(pk["seed_rbeg"], res.seeds["rbeg"])
    e3.close()
    e2 = ca.Engine(world["ix"], 0, fused=1)                                 # second, independent implementation
    f = e2.seed_batch(hb[: 60000 * 150], ho[: 60001])
    e2.close()
    k = int(res.mem_off[60000])
    assert np.array_equal(f.mems, res.mems[:k]) and np.array_equal(f.seeds, res.seeds[: int(res.seed_off[60000])])


def test_sample_is_bit_exact_vs_oracle(world):
    bw, sa = world["ix"].arrays()
    v = world["ix"].view
    o = _oracle.OracleIndex.from_arrays(v.primary, [v.L2[i] for i in range(1, 5)], bw, sa, 32)
    n = 30000
    want = o.seed_batch(world["bases"][: n * 150], world["off"][: n + 1], mode=1, threads=8)
    r = world["res"]
    assert np.array_equal(r.mem_off[: n + 1], want["mem_off"]) and np.array_equal(r.mems[: int(r.mem_off[n])], want["mems"])
    assert np.array_equal(r.seeds[: int(r.seed_off[n])], want["seeds"])


@pytest.mark.parametrize("kw", [dict(k=15), dict(k=16, y=100), dict(k=17, r=1.0), dict(k=18, s=1), dict(k=19, c=50), dict(k=19, y=0),
                                dict(k=19, y=5, r=1.2), dict(k=20), dict(k=25, r=1.1), dict(k=12, y=8), dict(k=19, y=1), dict(k=19, y=2),
                                dict(k=17, r=1.0, y=20), dict(k=19, r=1.0, y=20)])
def test_parameter_sweep_shortcuts_on_off_and_oracle(world, kw):
    """every -k/-r/-y/-c/-s combination goes three ways: all shortcuts on (window scheme for 15 <= k <= 19, text arrays),
    everything off (`sst_mode=0`: the literal sweep), and the CPU oracle; reads include Ns and 1 % substitutions"""
    ca, eng = world["ca"], world["eng"]
    n = 40000
    hb, ho = world["bases"][: n * 150], world["off"][: n + 1]
    on = eng.seed_batch(hb, ho, ca.Params(**kw))
    off = eng.seed_batch(hb, ho, ca.Params(sst_mode=0, **kw))
    assert np.array_equal(on.mem_off, off.mem_off) and np.array_equal(on.mems, off.mems)
    assert np.array_equal(on.seed_off, off.seed_off) and np.array_equal(on.seeds, off.seeds)
    bw, sa = world["ix"].arrays()
    v = world["ix"].view
    o = _oracle.OracleIndex.from_arrays(v.primary, [v.L2[i] for i in range(1, 5)], bw, sa, 32)
    m = 8000
    want = o.seed_batch(hb[: m * 150], ho[: m + 1], _oracle.make_params(**kw), mode=1, threads=8)
    assert np.array_equal(on.mem_off[: m + 1], want["mem_off"]) and np.array_equal(on.mems[: int(on.mem_off[m])], want["mems"])
    assert np.array_equal(on.seeds[: int(on.seed_off[m])], want["seeds"])


def _long_reads(g, n, lo, hi, p_sub, p_n, seed):
    rng = np.random.default_rng(seed)
    reads = []
    for j in range(n):
        ln = int(rng.integers(lo, hi + 1)); st = int(rng.integers(0, g.size - ln))
        q = g[st: st + ln].copy()
        if j & 1:
            q = (3 - q[::-1]).astype(np.uint8)
        if p_sub > 0:
            mut = rng.random(ln) < p_sub
            q[mut] = (q[mut] + rng.integers(1, 4, int(mut.sum()))).astype(np.uint8) % 4
        if p_n > 0:
            q[rng.random(ln) < p_n] = 4
        reads.append(q)
    bases = np.frombuffer(b"ACGTN", np.uint8)[np.concatenate(reads)]
    off = np.concatenate([[0], np.cumsum([r.size for r in reads])]).astype(np.uint64)
    return bases, off


def _oracle_of(world):
    bw, sa = world["ix"].arrays()
    v = world["ix"].view
    return _oracle.OracleIndex.from_arrays(v.primary, [v.L2[i] for i in range(1, 5)], bw, sa, 32)


@pytest.mark.parametrize("kw", [dict(), dict(k=255), dict(k=300, y=5), dict(k=254, r=1.0)])
def test_long_reads_vs_oracle(world, kw):
    """40 reads of 5-30 kbp (1 % substitutions, a few Ns), far beyond the lengths the packed fields of the shortcuts were
    sized on: every path has to fall back cleanly where a field would overflow; result identical to the oracle.
    -k 254 / 255 / 300: min_seed_len + 1 meets the 255 cap of the rep[] / lcp[] bytes (round 3 must then stay on the index)"""
    ca, eng, g = world["ca"], world["eng"], world["g"]
    bases, off = _long_reads(g, 40, 5000, 30000, 0.003 if kw else 0.01, 0.0005, 99)
    got = eng.seed_batch(bases, off, ca.Params(**kw))
    o = _oracle_of(world)
    want = o.seed_batch(bases, off, _oracle.make_params(**kw), mode=1, threads=8)
    assert want["stats"]["n_mems"] > 0
    assert np.array_equal(got.mem_off, want["mem_off"]) and np.array_equal(got.mems, want["mems"])
    assert np.array_equal(got.seed_off, want["seed_off"]) and np.array_equal(got.seeds, want["seeds"])


@pytest.mark.parametrize("sst", [1, 0])
def test_many_exact_long_reads_overflow_records(world, sst):
    """300 error-free reads of 10 kbp: ~500 round-3 seeds per read against 64 slots, so round 3 fills the overflow records
    (nb * 4 + 65536) on its side stream after the task loop has ended; the engine must notice, redo the batch with the
    fused kernel and still return the oracle's result (the flag used to be read only inside the loop)"""
    ca, eng, g = world["ca"], world["eng"], world["g"]
    bases, off = _long_reads(g, 300, 10000, 10000, 0.0, 0.0, 7)
    eng.reset_stats()
    got = eng.seed_batch(bases, off, ca.Params(sst_mode=sst))
    st = eng.stats()
    assert st["overflow_kernel_launches"] > 0 or st["overflow_mems"] > 0
    o = _oracle_of(world)
    want = o.seed_batch(bases, off, mode=1, threads=8)
    assert np.array_equal(got.mem_off, want["mem_off"]) and np.array_equal(got.mems, want["mems"])
    assert np.array_equal(got.seed_off, want["seed_off"]) and np.array_equal(got.seeds, want["seeds"])
    assert int(np.diff(want["mem_off"].astype(np.int64)).max()) > 64


def test_repeat_rich_and_indel_workloads_at_500mbp():
    """The unfriendly end of the workload range (tools/synth.py profiles; a real hg19 is ~50 % repeats, mostly old, diverged copies) in the
    driver-run tests: a 500 Mbp genome with five interspersed families at 2-25 % divergence and satellite arrays (`repeat50`), and 2 M reads each
    of three kinds -- 0.5 % substitutions, 1 % substitutions + 0.1 %/base indels, 2 % substitutions + indels.  For each: every shortcut on vs
    `sst_mode = 0` (the literal algorithm on the FM index) by device-side digests of all four result arrays over ALL reads, and a strided sample
    of 100,000 reads (every 20th) bit for bit against the oracle.  Reference: bwt_smem1a's sweep whose equal-size merging must survive
    (FM_index/bwt.c:325-345) and the re-seeding rule (mapping/bwamem.c:241-249), which these reads exercise on every call."""
    import torch
    import compseed_amd as ca
    import synth
    n = 2_000_000
    G = synth.make_genome(500_000_000, seed=20261004, device="cuda", **synth.PROFILES["repeat50"]["genome"])
    ix = ca.Index.build(G.cpu().numpy(), 0)
    eng = ca.Engine(ix, 0)
    chk = eng.check_index(G.data_ptr(), G.numel())                                 # the repeat-rich index itself (deep LCPs: satellites, young families)
    assert chk["text_checked"] == 1 and all(chk[k] == 0 for k in ("order_violations", "isa_violations", "bwt_violations", "sampled_sa_violations", "undecided_rows", "text_violations")), chk
    bw, sa = ix.arrays()
    v = ix.view
    o = _oracle.OracleIndex.from_arrays(v.primary, [v.L2[i] for i in range(1, 5)], bw, sa, 32)
    ids = np.arange(0, n, 20, dtype=np.uint64)
    sel = (ids[:, None].astype(np.int64) * 150 + np.arange(150)[None, :]).reshape(-1)
    ho = (np.arange(ids.size + 1, dtype=np.uint64) * np.uint64(150))
    wide = 0
    for k, rkw in enumerate((dict(p_sub=0.005), dict(p_sub=0.01, p_indel=0.001), dict(p_sub=0.02, p_indel=0.001))):
        bases, off = synth.make_reads(G, n, 150, seed=900 + k, sort=True, **rkw)
        hb = bases[torch.from_numpy(sel).to(bases.device)].cpu().numpy()
        torch.cuda.synchronize()
        eng.reset_stats()
        eng.seed_batch_device(bases.data_ptr(), off.data_ptr(), n, bases.numel(), ca.Params())
        d_on = eng.result_digest()
        st = eng.stats()
        assert st["reseed_text_calls"] > 0 and st["r3_text_seeds"] > 0 and st["sweep_text_calls"] > 0 and st["bwt_calls"] < st["bwt_queries"]
        got = eng.gather_reads(ids)
        want = o.seed_batch(hb, ho, _oracle.make_params(), mode=1, threads=16)
        assert np.array_equal(got.mem_off, want["mem_off"]) and np.array_equal(got.mems, want["mems"]), rkw
        assert np.array_equal(got.seed_off, want["seed_off"]) and np.array_equal(got.seeds, want["seeds"]), rkw
        wide += int((np.diff(want["mem_off"].astype(np.int64)) > 64).sum())               # reads with more mems than the first-pass arena holds per read
        eng.reset_stats()
        eng.seed_batch_device(bases.data_ptr(), off.data_ptr(), n, bases.numel(), ca.Params(sst_mode=0))
        st0 = eng.stats()
        assert st0["bwt_calls"] == st0["bwt_queries"] and st0["reseed_text_calls"] == 0
        assert eng.result_digest() == d_on, rkw                                    # all 2 M reads, all four arrays
        del bases, off
    assert wide > 0                                                                # ... occurred: the overflow records and the wave-per-read sort were exercised
    o.close(); eng.close(); ix.close()


def test_full_baseline_size_properties():
    """BASELINE configs[1] shape: hg19-size index (rows > 2^32: the 64-bit suffix-array / inverse-SA instantiation), 10 M x 150 bp
    reads.  For the default parameters and for configs[4]'s aggressive re-seeding (-r 1.0 -y 20):
      * every shortcut on vs `sst_mode = 0` (the literal algorithm on the FM index): device-side digests of all four result
        arrays over all 10 M reads must be equal -- no 6 GB download;
      * a strided sample of 100,000 reads (every 100th, so the whole genome incl. its repeats is covered) bit-exact vs the oracle;
      * cheap size-independent checks streamed through the host: counts, sortedness, filters; two runs give the same digest."""
    import torch
    import compseed_amd as ca
    import synth
    n = 10_000_000
    G = synth.make_genome(3_100_000_000, seed=20261003, device="cuda")
    ix = ca.Index.build(G.cpu().numpy(), 0)
    bases, off = synth.make_reads(G, n, 150, seed=777, p_sub=0.005, sort=True)
    # two more chunks of the same run for the streamed part at the end (BASELINE configs[2])
    extra = [synth.make_reads(G, n, 150, seed=778 + j, p_sub=0.005, sort=True, lo_frac=0.3 * (j + 1), hi_frac=0.3 * (j + 1) + 0.3)[0] for j in range(2)]
    assert ix.view.seq_len + 1 > 2**32
    eng = ca.Engine(ix, 0)
    # the index at the size it is used, checked independently of how it was built: recovered text == G + revcomp(G), all 6.2e9 neighbouring
    # suffix pairs in order (compared on the text), ISA o SA = id, BWT characters, sampled SA (cs_engine_check_index)
    chk = eng.check_index(G.data_ptr(), G.numel())
    assert chk["rows_checked"] == 2 * G.numel() and chk["text_checked"] == 1
    assert all(chk[k] == 0 for k in ("order_violations", "isa_violations", "bwt_violations", "sampled_sa_violations", "undecided_rows", "text_violations")), chk
    del G
    torch.cuda.empty_cache()
    bw, sa = ix.arrays()
    v = ix.view
    o = _oracle.OracleIndex.from_arrays(v.primary, [v.L2[i] for i in range(1, 5)], bw, sa, 32)
    ids = np.arange(0, n, n // 100000, dtype=np.uint64)[:100000]               # every 100th read: 100,000 reads against the oracle
    sel = (ids[:, None].astype(np.int64) * 150 + np.arange(150)[None, :]).reshape(-1)
    hb = bases[torch.from_numpy(sel).to(bases.device)].cpu().numpy()
    ho = (np.arange(ids.size + 1, dtype=np.uint64) * np.uint64(150))
    torch.cuda.synchronize()
    for kw in (dict(), dict(r=1.0, y=20)):
        r = eng.seed_batch_device(bases.data_ptr(), off.data_ptr(), n, bases.numel(), ca.Params(**kw))
        d_on = eng.result_digest()
        st = eng.stats()
        assert st["reseed_text_calls"] > 0 and st["r3_text_seeds"] > 0 and st["sweep_text_calls"] > 0      # the shortcuts did run
        got = eng.gather_reads(ids)
        want = o.seed_batch(hb, ho, _oracle.make_params(**kw), mode=1, threads=16)
        assert np.array_equal(got.mem_off, want["mem_off"]) and np.array_equal(got.mems, want["mems"]), kw
        assert np.array_equal(got.seed_off, want["seed_off"]) and np.array_equal(got.seeds, want["seeds"]), kw
        if not kw:  # stream the mems through the host in pieces: filters and sortedness over all 10 M reads
            mo = eng.download(r.ptr["mem_off"], np.uint64, n + 1)
            assert int(mo[-1]) == r.n_mems and (np.diff(mo.astype(np.int64)) >= 0).all() and r.n_mems > 50_000_000
            first = np.zeros(r.n_mems + 1, dtype=bool); first[mo.astype(np.int64)] = True
            step = 4_000_000
            for s0 in range(0, r.n_mems, step):
                k = min(step, r.n_mems - s0)
                m = eng.download(r.ptr["mems"] + s0 * 32, ca.INTV_DT, k)
                beg = (m["info"] >> np.uint64(32)).astype(np.int64); end = (m["info"] & np.uint64(0xffffffff)).astype(np.int64)
                assert (end - beg >= 19).all() and (end <= 150).all() and (m["x2"] >= 1).all()
                inside = ~first[s0 + 1: s0 + k]                                   # consecutive mems of the same read
                assert (m["info"][1:][inside] >= m["info"][:-1][inside]).all()    # sorted by info (comp_seed.cpp:2301)
            del first
        eng.seed_batch_device(bases.data_ptr(), off.data_ptr(), n, bases.numel(), ca.Params(**kw))
        assert eng.result_digest() == d_on                                         # deterministic despite atomics / task order
        # the same batch twice as a stream of device batches: both pass contexts run it AT THE SAME TIME and each must deliver the same 10 M results
        eng.submit_device(bases.data_ptr(), off.data_ptr(), n, bases.numel(), ca.Params(**kw))
        eng.submit_device(bases.data_ptr(), off.data_ptr(), n, bases.numel(), ca.Params(**kw))
        eng.collect_device()
        eng.submit_device(bases.data_ptr(), off.data_ptr(), n, bases.numel(), ca.Params(**kw))      # a third one beside the second: the first context again
        eng.collect_device()
        d1 = None
        eng.collect_device(); d1 = eng.result_digest()                              # (the last result: held by the first context)
        assert d1 == d_on, kw
        eng.submit_device(bases.data_ptr(), off.data_ptr(), n, bases.numel(), ca.Params(**kw))
        eng.submit_device(bases.data_ptr(), off.data_ptr(), n, bases.numel(), ca.Params(**kw))
        eng.collect_device(); eng.collect_device()
        assert eng.result_digest() == d_on, kw                                      # ... and by the second
        eng.reset_stats()
        eng.seed_batch_device(bases.data_ptr(), off.data_ptr(), n, bases.numel(), ca.Params(sst_mode=0, **kw))
        st0 = eng.stats()
        assert st0["reseed_text_calls"] == 0 and st0["r3_text_seeds"] == 0 and st0["bwt_calls"] == st0["bwt_queries"]
        assert eng.result_digest() == d_on, kw                                     # all 10 M reads, all four arrays
        eng.reset_stats()
    # ---- BASELINE configs[2] shape: ~600 M reads streamed through ONE GPU as 60 chunks of 10 M, three chunks in flight (cs_engine_submit /
    # cs_engine_collect_packed), the chunks coming from pinned host buffers as a reader would deliver them.  Three distinct chunks are cycled
    # (generating 90 GB of distinct text would only time the generator); every collected chunk must be exactly the result the device
    # variant gives for that chunk: counts, offsets, and the packed mems / seeds of a strided sample.
    ho = off.cpu().numpy().astype(np.uint64)
    chunks = []
    for t in [bases] + extra:
        pin = ca.pinned_array(t.numel()); pin[:] = t.cpu().numpy()
        r = eng.seed_batch_device(t.data_ptr(), off.data_ptr(), n, t.numel())
        mo = eng.download(r.ptr["mem_off"], np.uint64, n + 1)
        g = eng.gather_reads(ids)
        chunks.append(dict(pin=pin, n_mems=r.n_mems, n_seeds=r.n_seeds, mo_sum=int(mo.sum(dtype=np.uint64)), mems=g.mems.copy(), rbeg=g.seeds["rbeg"].copy(), mem_off=mo))
    del extra
    assert len({c["n_mems"] for c in chunks}) == 3                                # the three chunks really differ
    import time
    n_chunks, done = 60, 0
    eng.submit(chunks[0]["pin"], ho); eng.submit(chunks[1]["pin"], ho); eng.submit(chunks[2]["pin"], ho)   # three in flight
    t0 = time.perf_counter()
    for i in range(n_chunks):
        p = eng.collect_packed()
        c = chunks[i % 3]
        assert (p["n_reads"], p["n_mems"], p["n_seeds"]) == (n, c["n_mems"], c["n_seeds"]), i
        if i % 7 == 0:                                                             # full offsets + a strided sample of mems and seeds
            assert int(p["mem_off"].sum(dtype=np.uint64)) == c["mo_sum"]
            sel_m = np.concatenate([np.arange(int(p["mem_off"][r]), int(p["mem_off"][r + 1])) for r in ids.astype(np.int64)])
            sel_s = np.concatenate([np.arange(int(p["seed_off"][r]), int(p["seed_off"][r + 1])) for r in ids.astype(np.int64)])
            assert np.array_equal(ca.unpack_mems16(p["mems"][sel_m]), c["mems"]) and np.array_equal(ca.packed_rbeg(p, sel_s), c["rbeg"]), i
        if i + 3 < n_chunks:
            eng.submit(chunks[(i + 3) % 3]["pin"], ho)
        done += n
    dt = time.perf_counter() - t0
    print("streamed %d reads in %.2f s: %.1f M reads/s incl. PCIe" % (done, dt, done / dt / 1e6))
    assert done == 600_000_000
    o.close(); eng.close(); ix.close()
