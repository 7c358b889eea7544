"""ctypes binding of oracle/libcsoracle.so -- the CPU restatement used as the CHECKER by the tests.

Test infrastructure only: nothing under compseed_amd/ imports this module or the library behind it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libcsoracle.so")

INTV_DT = np.dtype([("x0", "<u8"), ("x1", "<u8"), ("x2", "<u8"), ("info", "<u8")])
SEED_DT = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4")])


class Intv(C.Structure):
    _fields_ = [("x0", C.c_uint64), ("x1", C.c_uint64), ("x2", C.c_uint64), ("info", C.c_uint64)]


class Index(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64), ("bwt_size", C.c_uint64),
                ("bwt", C.c_void_p), ("sa_intv", C.c_uint64), ("n_sa", C.c_uint64), ("sa", C.c_void_p),
                ("owned_bwt", C.c_void_p), ("owned_sa", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("min_seed_len", C.c_int32), ("split_factor", C.c_float), ("split_width", C.c_int32),
                ("max_occ", C.c_int32), ("max_mem_intv", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("bwt_queries", "bwt_calls", "bwt_blocks", "bwt_blocks_uncached", "sal_queries",
                                          "sal_calls", "sal_steps", "sal_steps_uncached", "n_mems", "n_seeds",
                                          "q_fwd", "q_bwd", "q_r3", "n_calls", "lep_sum", "lep_max", "bwd_steps")] + \
               [("lep_hist", C.c_uint64 * 8), ("q_bwd_hist", C.c_uint64 * 8)]

    def asdict(self):
        return {n: (list(getattr(self, n)) if n.endswith("hist") else int(getattr(self, n))) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        src_newer = (not os.path.exists(LIB_PATH) or
                     any(os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
                         for f in ("cs_oracle.c", "cs_oracle.h", "cs_index_naive.c", "cs_bsw_oracle.c")))
        if src_newer:
            subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, capture_output=True)
        L = C.CDLL(LIB_PATH)
        L.cso_index_load.argtypes = [C.POINTER(Index), C.c_char_p]
        L.cso_index_wrap.argtypes = [C.POINTER(Index), C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]
        L.cso_index_free.argtypes = [C.POINTER(Index)]
        L.cso_index_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(Index)]
        L.cso_occ4.argtypes = [C.POINTER(Index), C.c_uint64, C.POINTER(C.c_uint64)]
        L.cso_2occ4.argtypes = [C.POINTER(Index), C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.cso_extend.argtypes = [C.POINTER(Index), C.POINTER(Intv), C.POINTER(Intv), C.c_int]
        L.cso_sa.argtypes = [C.POINTER(Index), C.c_uint64, C.POINTER(C.c_uint64)]
        L.cso_sa.restype = C.c_uint64
        L.cso_params_default.argtypes = [C.POINTER(Params)]
        L.cso_seed_batch.argtypes = [C.POINTER(Index), C.POINTER(Params), C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_void_p), C.POINTER(Stats)]
        L.cso_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def make_params(k=19, r=1.5, s=10, c=500, y=20):
    p = Params()
    p.min_seed_len, p.split_factor, p.split_width, p.max_occ, p.max_mem_intv = k, r, s, c, y
    return p


class OracleIndex:
    """Index handle: from files (`prefix`) or from arrays (`from_arrays`)."""

    def __init__(self, prefix=None):
        self.idx = Index()
        self._keep = []
        if prefix is not None:
            rc = lib().cso_index_load(C.byref(self.idx), prefix.encode())
            if rc:
                raise IOError("cso_index_load(%s) failed: %d" % (prefix, rc))

    @classmethod
    def from_arrays(cls, primary, L2_1to4, bwt_words, sa, sa_intv=32):
        self = cls()
        l2 = np.ascontiguousarray(L2_1to4, dtype=np.uint64)
        bwt = np.ascontiguousarray(bwt_words, dtype=np.uint32)
        sa = np.ascontiguousarray(sa, dtype=np.uint64)
        self._keep = [l2, bwt, sa]
        lib().cso_index_wrap(C.byref(self.idx), int(primary), l2.ctypes.data, bwt.ctypes.data, bwt.size, sa.ctypes.data, sa.size, sa_intv)
        return self

    @classmethod
    def build(cls, fwd_nt4, threads=8):
        """index of a forward-strand genome (codes 0..3) by the oracle's naive suffix sort (cs_index_naive.c)"""
        self = cls()
        g = np.ascontiguousarray(fwd_nt4, dtype=np.uint8)
        rc = lib().cso_index_build(g.ctypes.data, g.size, threads, C.byref(self.idx))
        if rc:
            raise RuntimeError("cso_index_build failed: %d" % rc)
        return self

    def arrays(self):
        """(bwt words, sampled SA) as numpy copies"""
        i = self.idx
        bw = np.frombuffer((C.c_char * (int(i.bwt_size) * 4)).from_address(i.bwt), dtype="<u4").copy()
        sa = np.frombuffer((C.c_char * (int(i.n_sa) * 8)).from_address(i.sa), dtype="<u8").copy()
        return bw, sa

    def close(self):
        lib().cso_index_free(C.byref(self.idx))

    def occ4(self, k):
        out = (C.c_uint64 * 4)()
        lib().cso_occ4(C.byref(self.idx), C.c_uint64(int(k) & (2**64 - 1)), out)
        return list(out)

    def occ2x4(self, k, l):
        a = (C.c_uint64 * 4)(); b = (C.c_uint64 * 4)()
        nb = lib().cso_2occ4(C.byref(self.idx), C.c_uint64(int(k) & (2**64 - 1)), C.c_uint64(int(l) & (2**64 - 1)), a, b)
        return list(a), list(b), nb

    def extend(self, x0, x1, x2, is_back):
        ik = Intv(int(x0), int(x1), int(x2), 0)
        ok = (Intv * 4)()
        lib().cso_extend(C.byref(self.idx), C.byref(ik), ok, int(is_back))
        return [(o.x0, o.x1, o.x2) for o in ok]

    def sa(self, k):
        return int(lib().cso_sa(C.byref(self.idx), int(k), None))

    def seed_batch(self, bases, offsets, params=None, mode=0, sst_batch=512, want_sal=True, threads=1):
        """bases: uint8 array (ASCII or nt4), offsets: uint64[n+1].  Returns dict of numpy arrays + stats."""
        params = params or make_params()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        mo, mm, so, ss = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        st = Stats()
        rc = lib().cso_seed_batch(C.byref(self.idx), C.byref(params), n, bases.ctypes.data, offsets.ctypes.data, mode, sst_batch,
                                  int(want_sal), threads, C.byref(mo), C.byref(mm), C.byref(so), C.byref(ss), C.byref(st))
        assert rc == 0
        def take(ptr, dt, cnt):
            if cnt == 0:
                return np.zeros(0, dtype=dt)
            buf = (C.c_char * (cnt * np.dtype(dt).itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dt, count=cnt).copy()
        out = dict(mem_off=take(mo, "<u8", n + 1), mems=take(mm, INTV_DT, int(st.n_mems)),
                   seed_off=take(so, "<u8", n + 1), seeds=take(ss, SEED_DT, int(st.n_seeds)), stats=st.asdict())
        for p in (mo, mm, so, ss):
            lib().cso_free(p)
        return out


# ---------------------------------------------------------------------------------------------------------------------
# banded Smith-Waterman seed extension (oracle/cs_bsw_oracle.c) and its fixtures (tests/golden/bsw1/, from the real reference)
BSW_PAIR_DT = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"), ("h0", "<i4"), ("pad", "<i4")])
BSW_RES_DT = np.dtype([("score", "<i4"), ("qle", "<i4"), ("tle", "<i4"), ("gtle", "<i4"), ("gscore", "<i4"), ("max_off", "<i4")])
BSW_DIR = os.path.join(ROOT, "tests", "golden", "bsw1")


class BswParams(C.Structure):
    _fields_ = [("mat", C.c_int8 * 25), ("o_del", C.c_int32), ("e_del", C.c_int32), ("o_ins", C.c_int32), ("e_ins", C.c_int32),
                ("zdrop", C.c_int32), ("end_bonus", C.c_int32)]


def bsw_fixture(tag):
    """tests/golden/bsw1/<tag>.bsw.npz -> dict(mat, meta [n,17] = kind w zdrop end_bonus o_del e_del o_ins e_ins qlen tlen h0 | score qle tle gtle
    gscore max_off, pairs (BSW_PAIR_DT), qbuf, tbuf, want (BSW_RES_DT))"""
    z = np.load(os.path.join(BSW_DIR, tag + ".bsw.npz"))
    meta = z["meta"]
    pairs = np.zeros(meta.shape[0], dtype=BSW_PAIR_DT)
    pairs["q_off"], pairs["t_off"] = z["q_off"][:-1], z["t_off"][:-1]
    pairs["qlen"], pairs["tlen"], pairs["h0"] = meta[:, 8], meta[:, 9], meta[:, 10]
    want = np.zeros(meta.shape[0], dtype=BSW_RES_DT)
    for i, n in enumerate(BSW_RES_DT.names):
        want[n] = meta[:, 11 + i]
    return dict(mat=z["mat"], meta=meta, pairs=pairs, qbuf=z["qbuf"].copy(), tbuf=z["tbuf"].copy(), want=want)


def bsw_groups(meta):
    """indices of the records grouped by (w, zdrop, end_bonus, o_del, e_del, o_ins, e_ins): one extender object / one call each"""
    keys = {}
    for i, row in enumerate(meta[:, 1:8]):
        keys.setdefault(tuple(int(x) for x in row), []).append(i)
    return {k: np.array(v, dtype=np.int64) for k, v in keys.items()}


def bsw_params(mat, key):
    P = BswParams()
    for i in range(25):
        P.mat[i] = int(mat[i])
    _, P.zdrop, P.end_bonus, P.o_del, P.e_del, P.o_ins, P.e_ins = key
    return P


def bsw_extend(fx, rule=None, threads=4):
    """the oracle on every record of a fixture; rule None = the reference's dispatch per pair, 0 = matrix (ksw_extend2), 1 = vector rule"""
    L = lib()
    L.cso_extend_batch.argtypes = [C.POINTER(BswParams), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.cso_extend_pair_rule.argtypes = [C.POINTER(BswParams), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    got = np.zeros(fx["meta"].shape[0], dtype=BSW_RES_DT)
    qb, tb = np.ascontiguousarray(fx["qbuf"]), np.ascontiguousarray(fx["tbuf"])
    if qb.size == 0:
        qb = np.zeros(1, np.uint8)
    if tb.size == 0:
        tb = np.zeros(1, np.uint8)
    for key, idx in bsw_groups(fx["meta"]).items():
        P = bsw_params(fx["mat"], key)
        if rule is None:
            pr = np.ascontiguousarray(fx["pairs"][idx]); out = np.zeros(idx.size, dtype=BSW_RES_DT)
            assert L.cso_extend_batch(C.byref(P), idx.size, pr.ctypes.data, qb.ctypes.data, tb.ctypes.data, key[0], threads, out.ctypes.data) == 0
            got[idx] = out
        else:
            one = np.zeros(1, dtype=BSW_RES_DT)
            for i in idx:
                p = fx["pairs"][i]
                assert L.cso_extend_pair_rule(C.byref(P), rule, int(p["qlen"]), qb.ctypes.data + int(p["q_off"]), int(p["tlen"]), tb.ctypes.data + int(p["t_off"]),
                                              key[0], int(p["h0"]), one.ctypes.data) == 0
                got[i] = one[0]
    return got
