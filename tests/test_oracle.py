"""The oracle (oracle/cs_oracle.c) pinned against the golden vectors produced by the real reference.

CPU-only.  Every golden run is replayed in both flows: mode 0 = uncached BWA-MEM control flow
(bwamem.c:218-272), mode 1 = CompSeed control flow with the emulated SST (comp_seed.cpp:2255-2347).
Mems, seeds, order and the reference's own counters must all be bit-identical.
"""
import os

import numpy as np
import pytest

import _data
import _oracle


@pytest.fixture(scope="module")
def oidx():
    ix = _oracle.OracleIndex(_data.PREFIX)
    yield ix
    ix.close()


def test_index_header(oidx):
    f = _data.load_bwt_files()
    assert oidx.idx.primary == f["primary"]
    assert list(oidx.idx.L2)[1:] == [int(x) for x in f["L2"]]
    assert oidx.idx.seq_len == f["seq_len"]
    assert oidx.idx.sa_intv == 32
    assert oidx.idx.n_sa == (f["seq_len"] + 32) // 32 == f["sa"].size
    assert oidx.idx.bwt_size == f["bwt"].size


def test_index_wrap_equals_load(oidx):
    f = _data.load_bwt_files()
    w = _oracle.OracleIndex.from_arrays(f["primary"], f["L2"], f["bwt"], f["sa"], f["sa_intv"])
    for k in (0, 1, f["primary"], f["seq_len"], 12345):
        assert w.occ4(k) == oidx.occ4(k)
        assert w.sa(k) == oidx.sa(k)


def test_primitives_known_answers(oidx):
    z = np.load(os.path.join(_data.GOLD, "prims.npz"))
    for row in z["occ4"]:
        assert oidx.occ4(int(row[0])) == [int(x) for x in row[1:5]]
    for row in z["occ2x4"]:
        a, b, _ = oidx.occ2x4(int(row[0]), int(row[1]))
        assert a == [int(x) for x in row[2:6]] and b == [int(x) for x in row[6:10]]
    for row in z["ext"]:
        got = oidx.extend(int(row[0]), int(row[1]), int(row[2]), int(row[3]))
        want = [tuple(int(x) for x in row[4 + 3 * c: 7 + 3 * c]) for c in range(4)]
        assert got == want
    for row in z["sa"]:
        assert oidx.sa(int(row[0])) == int(row[1])


@pytest.mark.parametrize("name,pname", _data.golden_runs())
@pytest.mark.parametrize("mode", [0, 1])
def test_golden_seeds(oidx, name, pname, mode):
    z, kw = _data.load_golden(name, pname)
    bases, off = _data.load_reads(name)
    got = oidx.seed_batch(bases, off, _oracle.make_params(**kw), mode=mode, sst_batch=512, want_sal=True, threads=1)
    assert np.array_equal(got["mem_off"], z["mem_off"])
    m = got["mems"]
    assert np.array_equal(np.stack([m["x0"], m["x1"], m["x2"], m["info"]], axis=1), z["mems"])
    assert np.array_equal(got["seed_off"], z["seed_off"])
    assert np.array_equal(got["seeds"]["rbeg"], z["seed_rbeg"])
    assert np.array_equal(got["seeds"]["qbeg"], z["seed_qbeg"])
    assert np.array_equal(got["seeds"]["len"], z["seed_len"])
    c = z["counters"]  # n_reads, n_mems, n_seeds, bwt_queries, bwt_calls, sal_queries, sal_calls, n_diff
    st = got["stats"]
    assert st["bwt_queries"] == int(c[3])        # CompSeed "BWT-extend queries" == bwamem "BWT-extend calls"
    assert st["sal_queries"] == int(c[5]) and st["sal_calls"] == int(c[6])
    if mode == 1:
        assert st["bwt_calls"] == int(c[4])      # real calls under the reference's 512-read SST policy
    else:
        assert st["bwt_calls"] == st["bwt_queries"]


def test_threads_and_batching_do_not_change_seeds(oidx):
    bases, off = _data.load_reads("sorted150")
    a = oidx.seed_batch(bases, off, mode=1, sst_batch=512, threads=1)
    b = oidx.seed_batch(bases, off, mode=1, sst_batch=64, threads=4)
    c = oidx.seed_batch(bases, off, mode=0, sst_batch=100, threads=3)
    for k in ("mem_off", "mems", "seed_off", "seeds"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k])
    # cache statistics do depend on the batching (SURVEY 8a note), results do not
    assert a["stats"]["bwt_queries"] == b["stats"]["bwt_queries"] == c["stats"]["bwt_queries"]
    assert b["stats"]["bwt_calls"] >= a["stats"]["bwt_calls"]


def test_nt4_input_equals_ascii(oidx):
    bases, off = _data.load_reads("ragged")
    tbl = np.full(256, 4, dtype=np.uint8)
    for ch, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        tbl[ch] = v
    tbl[ord("-")] = 5
    a = oidx.seed_batch(bases, off)
    b = oidx.seed_batch(tbl[bases], off)
    for k in ("mem_off", "mems", "seed_off", "seeds"):
        assert np.array_equal(a[k], b[k])


def test_naive_index_builder_reproduces_the_bwaidx_fixture():
    """oracle/cs_index_naive.c (comparison sort of all suffixes) == the .bwt/.sa the reference's bwaidx wrote for the fixture"""
    f = _data.load_bwt_files()
    o = _oracle.OracleIndex.build(_data.load_pac_forward(), threads=4)
    bw, sa = o.arrays()
    assert o.idx.primary == f["primary"] and list(o.idx.L2)[1:] == [int(x) for x in f["L2"]]
    assert np.array_equal(bw, f["bwt"]) and np.array_equal(sa, f["sa"])
    o.close()


@pytest.fixture(scope="module")
def config1():
    """BASELINE.json configs[0]: E. coli K-12-size genome, 100 k x 100 bp reads (CPU plumbing + bit-exact seed diff)"""
    import hashlib
    import json
    want = json.load(open(os.path.join(_data.HERE, "golden", "c1", "config1.json")))
    ref, reads, shuf = _data.config1_dataset()
    assert hashlib.md5(ref.encode()).hexdigest() == want["genome_md5"], "the generator no longer reproduces the pinned genome"
    o = _oracle.OracleIndex.build(_data.codes_of(ref), threads=8)
    yield dict(want=want, o=o, sets={"sorted": reads, "shuffled": shuf})
    o.close()


def test_config1_index_is_bwaidx_identical(config1):
    import hashlib
    o, want = config1["o"], config1["want"]
    bw, sa = o.arrays()
    i = o.idx
    hdr = np.array([i.primary] + list(i.L2)[1:], dtype="<u8").tobytes()
    assert hashlib.md5(hdr + bw.tobytes()).hexdigest() == want["index_md5"]["bwt"]                       # bwt_dump_bwt, bwt.c:385-394
    sa_hdr = np.array([i.primary] + list(i.L2)[1:] + [i.sa_intv, i.seq_len], dtype="<u8").tobytes()
    assert hashlib.md5(sa_hdr + sa[1:].tobytes()).hexdigest() == want["index_md5"]["sa"]                  # bwt_dump_sa, bwt.c:396-407


@pytest.mark.parametrize("order", ["sorted", "shuffled"])
def test_config1_counters_and_output_are_the_references(config1, order):
    """the reference's own run of this set: 38,109,585 bwt_extend queries, 28,527,263 (sorted) /
    29,553,396 (shuffled) real calls under the 512-read SST policy, 569,493 SAL queries -- and the complete output, by digest"""
    import hashlib
    w = config1["want"]["sets"][order]
    reads = config1["sets"][order]
    txt = ("\n".join(reads) + "\n").encode()
    assert hashlib.md5(txt).hexdigest() == w["reads_md5"]
    bases, off = _data.pack_reads([r.encode() for r in reads])
    got = config1["o"].seed_batch(bases, off, mode=1, sst_batch=512, want_sal=True, threads=1)  # one thread: SST batches as `CompSeed -t 1`
    st = got["stats"]
    assert (st["bwt_queries"], st["bwt_calls"], st["sal_queries"], st["sal_calls"]) == (w["bwt_queries"], w["bwt_calls"], w["sal_queries"], w["sal_calls"])
    assert w["bwt_queries"] == 38109585 and w["sal_queries"] == 569493
    assert (st["n_mems"], st["n_seeds"]) == (w["n_mems"], w["n_seeds"])
    m = got["mems"]
    d = _data.digest_result(got["mem_off"], np.stack([m["x0"], m["x1"], m["x2"], m["info"]], axis=1), got["seed_off"],
                            got["seeds"]["rbeg"], got["seeds"]["qbeg"], got["seeds"]["len"])
    assert d == w["digest"]
