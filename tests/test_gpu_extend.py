"""Parity of the GPU banded Smith-Waterman seed extension (cs_extend_batch, compseed_amd/csrc/extend.hip) through the C ABI:
  * every extension the REAL reference performed on five read sets (recorded from its own run, tests/golden/bsw1/): all six outputs
    (score, qle, tle, gtle, gscore, max_off) of every pair bit-equal to what mapping/bandedSWA.cpp produced;
  * the known answers of the reference's scalar ksw_extend2 (bwalib/ksw.c:380): indels, narrow bands, Z-drop off / small, other gap penalties;
  * against the oracle (oracle/cs_bsw_oracle.c, itself pinned by the same fixtures) on fresh random pairs incl. queries of several hundred and
    several thousand bases (multi-chunk rows, the two-wave and one-wave LDS layouts, the HBM-scratch variant), and the device-pointer variant;
  * error behaviour: bad pairs are reported by code, the others are still delivered."""
import numpy as np
import pytest

import _oracle

pytestmark = pytest.mark.gpu

TRACES = ["main100.default", "sorted150.default", "ragged.default", "repeat100.default", "indel150.default", "indel150.scoring2"]


def _run_fixture(ca, fx, flags=0):
    got = np.zeros(fx["meta"].shape[0], dtype=ca.EXT_RES_DT)
    for key, idx in _oracle.bsw_groups(fx["meta"]).items():
        w, zdrop, end_bonus, o_del, e_del, o_ins, e_ins = key
        x = ca.Extender(0, ca.ExtParams(mat=fx["mat"], o_del=o_del, e_del=e_del, o_ins=o_ins, e_ins=e_ins, zdrop=zdrop, end_bonus=end_bonus, flags=flags))
        pr = np.zeros(idx.size, dtype=ca.EXT_PAIR_DT)
        for f in ("q_off", "t_off", "qlen", "tlen", "h0"):
            pr[f] = fx["pairs"][f][idx]
        got[idx] = x.extend(pr, fx["qbuf"], fx["tbuf"], w)
        x.close()
    return got


@pytest.mark.parametrize("flags", [0, 1, 2, 4, 8], ids=["default (lane per pair)", "packed16 above 64 bases", "packed16 for all", "wave per pair only", "lane per pair, base in the cell"])   # CS_EXT_*: every kernel and every mix of them must give the reference's numbers
@pytest.mark.parametrize("tag", TRACES)
def test_every_extension_of_the_reference_run(tag, flags):
    import compseed_amd as ca
    fx = _oracle.bsw_fixture(tag)
    got = _run_fixture(ca, fx, flags)
    for f in ca.EXT_RES_DT.names:
        assert np.array_equal(got[f], fx["want"][f]), (tag, f, int((got[f] != fx["want"][f]).sum()))


def test_ksw_extend2_known_answers():
    """generated pairs under 20 parameter sets x 7 band widths (asymmetric gap opens, gap extensions of 2 and 3, Z-drop off or small); no code
    above 4 occurs, so the vectorised scoring rule the kernel applies to short pairs and the matrix of ksw_extend2 are the same function"""
    import compseed_amd as ca
    for tag in ("kat_a1b4", "kat_a2b5"):
        fx = _oracle.bsw_fixture(tag)
        assert len(_oracle.bsw_groups(fx["meta"])) <= 160
        for flags in (0, 1, 2):
            got = _run_fixture(ca, fx, flags)
            for f in ca.EXT_RES_DT.names:
                assert np.array_equal(got[f], fx["want"][f]), (tag, flags, f, int((got[f] != fx["want"][f]).sum()))


def _random_pairs(rng, n, qlo, qhi, p_sub=0.03, p_gap=0.01):
    qs, ts, pairs = [], [], np.zeros(n, dtype=_oracle.BSW_PAIR_DT)
    qo = to = 0
    for i in range(n):
        ql = int(rng.integers(qlo, qhi + 1))
        q = rng.integers(0, 4, ql).astype(np.uint8)
        t = []
        j = 0
        while j < ql:
            u = rng.random()
            if u < p_gap / 2:
                t += list(rng.integers(0, 4, int(rng.integers(1, 6))))
            elif u < p_gap:
                j += int(rng.integers(1, 6)); continue
            t.append(int(rng.integers(0, 4)) if rng.random() < p_sub else int(q[j]))
            j += 1
        t = np.array(t + list(rng.integers(0, 4, int(rng.integers(0, 80)))), dtype=np.uint8)
        if rng.random() < 0.1 and ql > 3:
            q[int(rng.integers(0, ql))] = 4
        qs.append(q); ts.append(t)
        pairs[i] = (qo, to, ql, t.size, int(rng.integers(1, 200)), 0)
        qo += ql; to += t.size
    return pairs, np.concatenate(qs), np.concatenate(ts)


@pytest.mark.parametrize("qlo,qhi,n,par", [(1, 64, 3000, {}), (60, 200, 2000, {}), (150, 700, 400, {}), (1000, 3500, 24, {}), (4000, 7000, 6, {}), (9000, 12000, 3, {}),
                                           (1, 300, 1500, dict(o_del=4, e_del=2, o_ins=9, e_ins=3, zdrop=25, end_bonus=11, a=2, b=3)),
                                           (1, 300, 1500, dict(zdrop=0))])
def test_against_the_oracle_on_random_pairs(qlo, qhi, n, par):
    import compseed_amd as ca
    rng = np.random.default_rng(qlo * 7919 + qhi)
    pairs, qbuf, tbuf = _random_pairs(rng, n, qlo, qhi)
    P = ca.ExtParams(**par)
    for w in (100, 13):
        fx = dict(mat=np.array(list(P.mat), dtype=np.int8), pairs=pairs, qbuf=qbuf, tbuf=tbuf,
                  meta=np.tile(np.array([[0, w, P.zdrop, P.end_bonus, P.o_del, P.e_del, P.o_ins, P.e_ins] + [0] * 9], dtype=np.int32), (n, 1)))
        want = _oracle.bsw_extend(fx, threads=8)
        pr = np.zeros(n, dtype=ca.EXT_PAIR_DT)
        for f in ("q_off", "t_off", "qlen", "tlen", "h0"):
            pr[f] = pairs[f]
        for flags in (0, 1, 2, 4, 8):                        # lane per pair (default), the int16 kernel for long queries / for all, wave per pair only, base in the cell
            P.flags = flags
            x = ca.Extender(0, P)
            got = x.extend(pr, qbuf, tbuf, w)
            st = x.stats()
            x.close()
            assert np.array_equal(got, want.astype(ca.EXT_RES_DT)), (qlo, qhi, w, flags, int((got != want.astype(ca.EXT_RES_DT)).sum()))
            assert st["pairs"] == n and st["cells"] > 0 and st["rows"] > 0


def test_score_width_classes_of_the_lane_kernel():
    """the lane kernel keeps 8-bit scores where h0 + qlen x match <= 255 and 16-bit ones above: perfect matches whose score ends exactly at
    250 .. 262 (both sides of the boundary), match scores 1 and 2, against the oracle; and queries of 158 .. 163 columns (the lane kernel
    takes up to 160, the wave-per-pair kernel the rest)"""
    import compseed_amd as ca
    rng = np.random.default_rng(77)
    for a, cases in ((1, [(h0, L) for L in (100, 149, 150, 158, 160, 161, 163) for h0 in range(250 - L, 263 - L) if h0 >= 0]),
                     (2, [(h0, L) for L in (60, 100, 120) for h0 in range(250 - 2 * L, 263 - 2 * L) if h0 >= 0])):
        n = len(cases)
        pairs = np.zeros(n, dtype=_oracle.BSW_PAIR_DT)
        qs, ts = [], []
        qo = to = 0
        for k, (h0, L) in enumerate(cases):
            q = rng.integers(0, 4, L, dtype=np.uint8)
            t = np.concatenate([q, rng.integers(0, 4, 40, dtype=np.uint8)])
            pairs[k] = (qo, to, L, t.size, h0, 0)
            qs.append(q); ts.append(t); qo += L; to += t.size
        qbuf, tbuf = np.concatenate(qs), np.concatenate(ts)
        mat = np.array([(-1 if (i == 4 or j == 4) else (a if i == j else -4)) for i in range(5) for j in range(5)], dtype=np.int8)
        P = ca.ExtParams(mat=mat)
        fx = dict(mat=mat, pairs=pairs, qbuf=qbuf, tbuf=tbuf, meta=np.tile(np.array([[0, 100, P.zdrop, P.end_bonus, P.o_del, P.e_del, P.o_ins, P.e_ins] + [0] * 9], dtype=np.int32), (n, 1)))
        want = _oracle.bsw_extend(fx, threads=2).astype(ca.EXT_RES_DT)
        assert (want["score"] >= 250).all() and (want["score"] > 255).any() and (want["score"] <= 255).any()
        pr = np.zeros(n, dtype=ca.EXT_PAIR_DT)
        for f in ("q_off", "t_off", "qlen", "tlen", "h0"):
            pr[f] = pairs[f]
        for flags in (0, 4, 8):
            P.flags = flags
            x = ca.Extender(0, P)
            got = x.extend(pr, qbuf, tbuf, 100)
            x.close()
            assert np.array_equal(got, want), (a, flags, int((got != want).sum()))


def test_device_variant_and_errors():
    import compseed_amd as ca
    fx = _oracle.bsw_fixture("sorted150.default")
    n = fx["pairs"].size
    pr = np.zeros(n, dtype=ca.EXT_PAIR_DT)
    for f in ("q_off", "t_off", "qlen", "tlen", "h0"):
        pr[f] = fx["pairs"][f]
    ix = ca.Index.load(__import__("_data").PREFIX)
    eng = ca.Engine(ix, 0)                                   # (only for its device-memory helpers)
    x = ca.Extender(0)
    d_p, d_q, d_t, d_o = eng.alloc(pr.nbytes), eng.alloc(fx["qbuf"].size), eng.alloc(fx["tbuf"].size), eng.alloc(n * 24)
    eng.upload(d_p, pr); eng.upload(d_q, fx["qbuf"]); eng.upload(d_t, fx["tbuf"])
    x.extend_device(d_p, n, d_q, fx["qbuf"].size, d_t, fx["tbuf"].size, d_o, 100)
    got = eng.download(d_o, ca.EXT_RES_DT, n)
    assert np.array_equal(got, fx["want"].astype(ca.EXT_RES_DT))
    for d in (d_p, d_q, d_t, d_o):
        eng.free(d)
    # bad pairs: offsets beyond the buffers, qlen 0, negative tlen -> CS_EINVAL, zero results for those, the others delivered
    bad = pr[:50].copy()
    bad["q_off"][3] = fx["qbuf"].size; bad["qlen"][7] = 0; bad["tlen"][11] = -1; bad["t_off"][13] = 2**40
    with pytest.raises(ca.CSError) as ei:
        x.extend(bad, fx["qbuf"], fx["tbuf"], 100)
    assert ei.value.code == -1 and "4 pair(s)" in str(ei.value)
    out = x.last_out
    ok = np.ones(50, bool); ok[[3, 7, 11, 13]] = False
    assert np.array_equal(out[ok], fx["want"][:50][ok].astype(ca.EXT_RES_DT)) and (out[~ok].view(np.int32) == 0).all()
    assert x.extend(pr[:0], fx["qbuf"], fx["tbuf"], 100).size == 0          # empty batch
    with pytest.raises(ca.CSError):
        ca.Extender(0, ca.ExtParams(e_del=0))
    x.close(); eng.close(); ix.close()
