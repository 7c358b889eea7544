"""ctypes binding of libcompseed_amd.so (C ABI: include/compseed_amd.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
INTV_DT = np.dtype([("x0", "<u8"), ("x1", "<u8"), ("x2", "<u8"), ("info", "<u8")])   # cs_intv_t == bwtintv_t
SEED_DT = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4")])               # cs_seed_t

# every symbol include/compseed_amd.h declares (tests check the library exports exactly these)
SYMBOLS = ["cs_last_error", "cs_version", "cs_params_default", "cs_index_load", "cs_index_view", "cs_index_free", "cs_index_build",
           "cs_index_build_flags", "cs_index_save", "cs_refseq_from_fasta", "cs_refseq_codes", "cs_refseq_save", "cs_refseq_free", "cs_index_build_fasta", "cs_reader_open", "cs_reader_next", "cs_reader_close",
           "cs_chainer_create", "cs_chainer_destroy", "cs_chain_params_default", "cs_chain_batch", "cs_flt_params_default", "cs_chain_filter",
           "cs_device_count", "cs_engine_options_default", "cs_engine_create", "cs_engine_create_opts", "cs_engine_destroy", "cs_engine_seed_batch",
           "cs_engine_seed_batch_device", "cs_engine_submit_device", "cs_engine_collect_device", "cs_engine_seed_batch_packed", "cs_engine_submit", "cs_engine_collect_packed", "cs_unpack_mem", "cs_mem_seed_count", "cs_host_alloc", "cs_host_free", "cs_pack_reads",
           "cs_engine_result_digest", "cs_engine_gather_reads", "cs_engine_traffic_model", "cs_engine_stats", "cs_engine_reset_stats", "cs_engine_occ4",
           "cs_engine_extend", "cs_engine_sa", "cs_engine_probe_random_lines", "cs_device_alloc", "cs_device_free", "cs_device_upload",
           "cs_device_download", "cs_device_sync", "cs_packed_seed_rbeg", "cs_engine_check_index",
           "cs_ext_params_default", "cs_extender_create", "cs_extender_destroy", "cs_extend_batch", "cs_extend_batch_device", "cs_extender_stats",
           "cs_extender_upload", "cs_extend_batch_resident", "cs_aln_params_default", "cs_aligner_create", "cs_aligner_destroy", "cs_extend_chains", "cs_dedup_params_default", "cs_dedup_regions", "cs_aligner_stats"]


class CSError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("compseed_amd error %d: %s" % (code, msg))
        self.code = code


class IndexView(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64), ("bwt_size", C.c_uint64),
                ("bwt", C.c_void_p), ("sa_intv", C.c_uint64), ("n_sa", C.c_uint64), ("sa", C.c_void_p)]


class Params(C.Structure):
    """cs_params_t; defaults = mem_opt_init (mapping/comp_seed.cpp:26-58)."""
    _fields_ = [("min_seed_len", C.c_int32), ("split_factor", C.c_float), ("split_width", C.c_int32),
                ("max_occ", C.c_int32), ("max_mem_intv", C.c_uint64), ("want_sal", C.c_int32), ("sst_mode", C.c_int32),
                ("disable", C.c_uint32), ("count_traffic", C.c_uint32)]

    def __init__(self, k=19, r=1.5, s=10, c=500, y=20, want_sal=1, sst_mode=1, disable=0, count_traffic=0):
        super().__init__(k, r, s, c, y, want_sal, sst_mode, disable, count_traffic)


# cs_params_t.disable bits (include/compseed_amd.h CS_DISABLE_*)
DISABLE = dict(text_mode=0x01, r2_text=0x02, text_sweep=0x04, window=0x08, r3_text=0x10, kmer_filter=0x20, fwd0=0x40)


def disable_mask(*names):
    m = 0
    for n in names:
        m |= DISABLE[n]
    return m


class EngineOptions(C.Structure):
    """cs_engine_options_t; defaults come from cs_engine_options_default()."""
    _fields_ = [("full_sa", C.c_int32), ("sa64", C.c_int32), ("text_mode", C.c_int32), ("text_arrays", C.c_int32),
                ("jump_k", C.c_int32), ("kmer_filter", C.c_int32), ("fused", C.c_int32), ("mem_cap", C.c_int32),
                ("lep_arena_mb", C.c_int64), ("max_raw_mb", C.c_int64), ("r3_text_iter", C.c_int32),
                ("pipeline_reads", C.c_int32), ("expand_threads", C.c_int32),
                ("count_sal_merged", C.c_int32), ("verbose", C.c_int32), ("host_pack_threads", C.c_int32), ("passes_in_flight", C.c_int32), ("reserved", C.c_int32 * 3)]

    def __init__(self, **kw):
        super().__init__()
        load_library().cs_engine_options_default(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_) or k == "reserved":
                raise TypeError("unknown engine option %r" % k)
            setattr(self, k, v)


class CResult(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_mems", C.c_uint64), ("n_seeds", C.c_uint64), ("mem_off", C.c_void_p),
                ("mems", C.c_void_p), ("seed_off", C.c_void_p), ("seeds", C.c_void_p)]


class ChainParams(C.Structure):
    _fields_ = [("w", C.c_int32), ("max_chain_gap", C.c_int32), ("min_seed_len", C.c_int32), ("max_occ", C.c_int32)]

    def __init__(self, w=100, max_chain_gap=10000, k=19, c=500):
        super().__init__(w, max_chain_gap, k, c)


class FltParams(C.Structure):
    """cs_flt_params_t; defaults = mem_opt_init (mapping/comp_seed.cpp:26-58)"""
    _fields_ = [("min_chain_weight", C.c_int32), ("max_chain_extend", C.c_int32), ("max_chain_gap", C.c_int32), ("min_seed_len", C.c_int32),
                ("mask_level", C.c_float), ("drop_ratio", C.c_float), ("a", C.c_int32), ("b", C.c_int32), ("o_del", C.c_int32), ("e_del", C.c_int32),
                ("o_ins", C.c_int32), ("e_ins", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        load_library().cs_flt_params_default(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError("no such field: " + k)
            setattr(self, k, v)


class DedupParams(C.Structure):
    """cs_dedup_params_t; defaults = mem_opt_init (mapping/comp_seed.cpp:26-58)"""
    _fields_ = [("max_chain_gap", C.c_int32), ("mask_level_redun", C.c_float)]

    def __init__(self, max_chain_gap=10000, mask_level_redun=0.95):
        super().__init__(max_chain_gap, mask_level_redun)


class CChainResult(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_chains", C.c_uint64), ("n_seeds", C.c_uint64), ("chain_off", C.c_void_p), ("chains", C.c_void_p),
                ("cseed_off", C.c_void_p), ("cseeds", C.c_void_p)]


CHAIN_DT = np.dtype([("pos", "<i8"), ("rid", "<i4"), ("n_seeds", "<i4"), ("frac_rep", "<f4"), ("is_alt", "<i4")])   # cs_chain_t


class CPacked(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_mems", C.c_uint64), ("n_seeds", C.c_uint64), ("mem_format", C.c_int32), ("max_occ", C.c_int32),
                ("mem_off", C.c_void_p), ("mems", C.c_void_p), ("seed_off", C.c_void_p), ("seed_format", C.c_int32), ("reserved", C.c_int32),
                ("seed_rbeg_lo", C.c_void_p), ("seed_rbeg_hi", C.c_void_p)]


MEM16_DT = np.dtype([("w0", "<u8"), ("w1", "<u8")])   # cs_mem16_t


def unpack_mems16(p):
    """numpy restatement of cs_unpack_mem for CS_MEM_PACKED16 records -> INTV_DT array"""
    out = np.zeros(p.size, dtype=INTV_DT)
    m33 = np.uint64((1 << 33) - 1)
    out["x0"] = p["w0"] & m33
    out["x1"] = p["w1"] & m33
    out["x2"] = (p["w0"] >> np.uint64(33)) | ((p["w1"] >> np.uint64(63)) << np.uint64(31))
    out["info"] = (((p["w1"] >> np.uint64(33)) & np.uint64(0x7fff)) << np.uint64(32)) | ((p["w1"] >> np.uint64(48)) & np.uint64(0x7fff))
    return out


EXT_PAIR_DT = np.dtype([("q_off", "<u8"), ("t_off", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"), ("h0", "<i4"), ("reserved", "<i4")])   # cs_ext_pair_t
EXT_RES_DT = np.dtype([("score", "<i4"), ("qle", "<i4"), ("tle", "<i4"), ("gtle", "<i4"), ("gscore", "<i4"), ("max_off", "<i4")])       # cs_ext_result_t


class ExtParams(C.Structure):
    """cs_ext_params_t: the arguments of the reference's BandedPairWiseSW constructor (mapping/bandedSWA.h:117-121)"""
    _fields_ = [("mat", C.c_int8 * 25), ("o_del", C.c_int32), ("e_del", C.c_int32), ("o_ins", C.c_int32), ("e_ins", C.c_int32),
                ("zdrop", C.c_int32), ("end_bonus", C.c_int32), ("flags", C.c_uint32)]

    def __init__(self, mat=None, o_del=6, e_del=1, o_ins=6, e_ins=1, zdrop=100, end_bonus=5, a=1, b=4, flags=0):
        super().__init__()
        if mat is None:  # bwa_fill_scmat (bwalib/bwa.c:17-29)
            mat = [(-1 if (i == 4 or j == 4) else a if i == j else -b) for i in range(5) for j in range(5)]
        for i in range(25):
            self.mat[i] = int(mat[i])
        self.o_del, self.e_del, self.o_ins, self.e_ins, self.zdrop, self.end_bonus, self.flags = o_del, e_del, o_ins, e_ins, zdrop, end_bonus, flags


ALNREG_DT = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"), ("truesc", "<i4"), ("w", "<i4"),
                      ("seedcov", "<i4"), ("seedlen0", "<i4"), ("frac_rep", "<f4"), ("chain", "<i4")])   # cs_alnreg_t


class AlnParams(C.Structure):
    """cs_aln_params_t: the mem_opt_t fields the extension stage reads (-A -B -O -E -L -w -d)"""
    _fields_ = [(n, C.c_int32) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_clip5", "pen_clip3", "w", "zdrop", "threads")] + [("flags", C.c_uint32)]

    def __init__(self, a=1, b=4, o_del=6, e_del=1, o_ins=6, e_ins=1, pen_clip5=5, pen_clip3=5, w=100, zdrop=100, threads=16, flags=0):
        super().__init__(a, b, o_del, e_del, o_ins, e_ins, pen_clip5, pen_clip3, w, zdrop, threads, flags)


class CAlnResult(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_regs", C.c_uint64), ("reg_off", C.c_void_p), ("regs", C.c_void_p)]


class AlnStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("reads", "regions", "pairs", "retries", "purged", "launches", "ext_cells")] + [("ext_kernel_ms", C.c_double)]


class IndexCheck(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rows_checked", "order_violations", "isa_violations", "bwt_violations", "sampled_sa_violations", "undecided_rows",
                                          "text_violations")] + [("text_checked", C.c_int32), ("reserved", C.c_int32)]


class ExtStats(C.Structure):
    _fields_ = [("pairs", C.c_uint64), ("cells", C.c_uint64), ("rows", C.c_uint64), ("launches", C.c_uint64), ("kernel_ms", C.c_double)]


class PackedResult(dict):
    """cs_packed_result_t as a dict of zero-copy views; "seed_rbeg" is the 40-bit positions expanded to int64 on first use
    (cs_packed_seed_rbeg for every seed); packed_rbeg(p, sel) expands a selection only"""

    def __missing__(self, key):
        if key == "seed_rbeg":
            v = None if self["seed_rbeg_lo"] is None else packed_rbeg(self, slice(None))
            self[key] = v
            return v
        raise KeyError(key)


def packed_rbeg(p, sel):
    return p["seed_rbeg_lo"][sel].astype(np.int64) | (p["seed_rbeg_hi"][sel].astype(np.int64) << 32)


class Stats(C.Structure):
    _fields_ = [("reads", C.c_uint64), ("bases", C.c_uint64), ("mems", C.c_uint64), ("seeds", C.c_uint64),
                ("bwt_queries", C.c_uint64), ("bwt_calls", C.c_uint64), ("sal_queries", C.c_uint64),
                ("sal_calls", C.c_uint64), ("overflow_mems", C.c_uint64), ("seed_kernel_ms", C.c_double),
                ("sal_kernel_ms", C.c_double), ("total_ms", C.c_double), ("seed_kernel_launches", C.c_uint64),
                ("overflow_kernel_ms", C.c_double), ("overflow_kernel_launches", C.c_uint64),
                ("reseed_text_calls", C.c_uint64), ("reseed_index_calls", C.c_uint64), ("sweep_text_calls", C.c_uint64),
                ("r3_text_seeds", C.c_uint64)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


KERNELS = ["fwd0_kernel", "fwd_kernel", "bwd_win_kernel", "bwd_win0_kernel", "bwd_wide_kernel", "bwd_all_kernel", "r2text_kernel",
           "r3text_kernel", "smem_kernel"]
EVENTS = ["occ_record", "jump_entry", "filter_word", "sa_entry", "isa_entry", "text_word", "rep_load", "lcp_byte", "lep_entry", "mem_record"]


class Traffic(C.Structure):
    _fields_ = [("events", (C.c_uint64 * 10) * 9), ("event_bytes", C.c_uint64 * 10), ("stream_bytes", C.c_uint64)]


class Digest(C.Structure):
    _fields_ = [("mem_off", C.c_uint64), ("mems", C.c_uint64), ("seed_off", C.c_uint64), ("seeds", C.c_uint64)]

    def astuple(self):
        return (self.mem_off, self.mems, self.seed_off, self.seeds)


def digest_words(a):
    """host restatement of cs_engine_result_digest for one array (any dtype, viewed as 64-bit words)"""
    w = np.ascontiguousarray(a).view(np.uint8).view("<u8").astype(np.uint64)
    with np.errstate(over="ignore"):
        z = w + np.arange(w.size, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return int(np.add.reduce(z, dtype=np.uint64)) if z.size else 0


def lib_path():
    # CS_LIB: load an alternative build of the same library (A/B experiments); the default is the in-tree build
    return os.environ.get("CS_LIB") or os.path.join(_HERE, "libcompseed_amd.so")


def build_library(force=False):
    """Compile the HIP library in-tree (hipcc --offload-arch=gfx950; works without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.run(["make", "-C", src, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", src, "all"], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("building libcompseed_amd.so failed:\n" + r.stdout + r.stderr)
    return lib_path()


_lib = None


def load_library():
    """Load the C-ABI library; fails loudly when it has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C compseed_amd/csrc`" % p)
    L = C.CDLL(p)
    vp, i64, u64p = C.c_void_p, C.c_int64, C.c_void_p
    L.cs_last_error.restype = C.c_char_p
    L.cs_version.restype = C.c_char_p
    L.cs_params_default.argtypes = [C.POINTER(Params)]
    L.cs_index_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.cs_index_view.argtypes = [vp, C.POINTER(IndexView)]
    L.cs_index_free.argtypes = [vp]
    L.cs_index_free.restype = None
    L.cs_index_build.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.cs_index_build_flags.argtypes = [vp, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(vp)]
    L.cs_engine_options_default.argtypes = [C.POINTER(EngineOptions)]
    L.cs_engine_options_default.restype = None
    L.cs_engine_create_opts.argtypes = [C.POINTER(IndexView), C.c_int, C.POINTER(EngineOptions), C.POINTER(vp)]
    L.cs_index_save.argtypes = [vp, C.c_char_p]
    L.cs_refseq_from_fasta.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.cs_refseq_codes.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.cs_refseq_save.argtypes = [vp, C.c_char_p]
    L.cs_refseq_free.argtypes = [vp]
    L.cs_refseq_free.restype = None
    L.cs_index_build_fasta.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.cs_reader_open.argtypes = [C.c_char_p, C.c_int64, C.POINTER(vp)]
    L.cs_reader_next.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64)]
    L.cs_reader_close.argtypes = [vp]
    L.cs_reader_close.restype = None
    L.cs_chainer_create.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.cs_chainer_destroy.argtypes = [vp]
    L.cs_chainer_destroy.restype = None
    L.cs_chain_params_default.argtypes = [C.POINTER(ChainParams)]
    L.cs_chain_params_default.restype = None
    L.cs_chain_batch.argtypes = [vp, C.POINTER(ChainParams), C.POINTER(CResult), vp, C.c_int, C.POINTER(CChainResult)]
    L.cs_flt_params_default.argtypes = [C.POINTER(FltParams)]
    L.cs_flt_params_default.restype = None
    L.cs_chain_filter.argtypes = [vp, C.POINTER(FltParams), C.POINTER(CChainResult), vp, vp, C.c_int, C.POINTER(CChainResult), C.POINTER(vp)]
    L.cs_device_count.argtypes = [C.POINTER(C.c_int)]
    L.cs_engine_create.argtypes = [C.POINTER(IndexView), C.c_int, C.POINTER(vp)]
    L.cs_engine_destroy.argtypes = [vp]
    L.cs_engine_destroy.restype = None
    L.cs_engine_seed_batch.argtypes = [vp, C.POINTER(Params), i64, vp, u64p, C.POINTER(CResult)]
    L.cs_engine_seed_batch_device.argtypes = [vp, C.POINTER(Params), i64, vp, u64p, C.c_uint64, C.POINTER(CResult)]
    L.cs_engine_submit_device.argtypes = [vp, C.POINTER(Params), i64, vp, vp, C.c_uint64]
    L.cs_engine_collect_device.argtypes = [vp, C.POINTER(CResult)]
    L.cs_engine_result_digest.argtypes = [vp, C.POINTER(Digest)]
    L.cs_engine_gather_reads.argtypes = [vp, i64, vp, C.POINTER(CResult)]
    L.cs_engine_traffic_model.argtypes = [vp, C.POINTER(Traffic)]
    L.cs_engine_seed_batch_packed.argtypes = [vp, C.POINTER(Params), i64, vp, u64p, C.POINTER(CPacked)]
    L.cs_engine_submit.argtypes = [vp, C.POINTER(Params), i64, vp, u64p]
    L.cs_engine_collect_packed.argtypes = [vp, C.POINTER(CPacked)]
    L.cs_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.cs_pack_reads.argtypes = [vp, vp, C.c_int64, vp, C.c_int, C.c_uint32]
    L.cs_host_free.argtypes = [vp]
    L.cs_engine_stats.argtypes = [vp, C.POINTER(Stats)]
    L.cs_engine_reset_stats.argtypes = [vp]
    L.cs_engine_reset_stats.restype = None
    L.cs_engine_occ4.argtypes = [vp, i64, vp, vp]
    L.cs_engine_extend.argtypes = [vp, i64, vp, vp, vp]
    L.cs_engine_sa.argtypes = [vp, i64, vp, vp]
    L.cs_engine_probe_random_lines.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.cs_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.cs_device_free.argtypes = [vp, vp]
    L.cs_device_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.cs_device_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.cs_device_sync.argtypes = [vp]
    L.cs_engine_check_index.argtypes = [vp, vp, C.c_uint64, C.POINTER(IndexCheck)]
    L.cs_ext_params_default.argtypes = [C.POINTER(ExtParams)]
    L.cs_ext_params_default.restype = None
    L.cs_extender_create.argtypes = [C.c_int, C.POINTER(ExtParams), C.POINTER(vp)]
    L.cs_extender_destroy.argtypes = [vp]
    L.cs_extender_destroy.restype = None
    L.cs_extend_batch.argtypes = [vp, i64, vp, vp, C.c_uint64, vp, C.c_uint64, C.c_int32, vp]
    L.cs_extend_batch_device.argtypes = [vp, i64, vp, vp, C.c_uint64, vp, C.c_uint64, C.c_int32, vp]
    L.cs_extender_stats.argtypes = [vp, C.POINTER(ExtStats)]
    L.cs_extender_upload.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
    L.cs_extend_batch_resident.argtypes = [vp, i64, vp, C.c_int32, vp]
    L.cs_aln_params_default.argtypes = [C.POINTER(AlnParams)]
    L.cs_aln_params_default.restype = None
    L.cs_aligner_create.argtypes = [C.c_char_p, C.c_int, C.POINTER(AlnParams), C.POINTER(vp)]
    L.cs_aligner_destroy.argtypes = [vp]
    L.cs_aligner_destroy.restype = None
    L.cs_extend_chains.argtypes = [vp, C.POINTER(CChainResult), vp, vp, vp, C.POINTER(CAlnResult)]
    L.cs_aligner_stats.argtypes = [vp, C.POINTER(AlnStats)]
    L.cs_dedup_params_default.argtypes = [C.POINTER(DedupParams)]
    L.cs_dedup_params_default.restype = None
    L.cs_dedup_regions.argtypes = [vp, C.POINTER(DedupParams), C.POINTER(CAlnResult), vp, vp, C.POINTER(CAlnResult), C.POINTER(vp)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise CSError(rc, load_library().cs_last_error().decode(errors="replace"))


class Index:
    """Host copy of an FM-index: from the reference's files (<prefix>.bwt/.sa) or from numpy arrays."""

    def __init__(self):
        self.view = IndexView()
        self._handle = None
        self._keep = []

    @classmethod
    def load(cls, prefix):
        self = cls()
        L = load_library()
        h = C.c_void_p()
        _check(L.cs_index_load(os.fsencode(prefix), C.byref(h)))
        self._handle = h
        _check(L.cs_index_view(h, C.byref(self.view)))
        return self

    @classmethod
    def from_arrays(cls, primary, L2_1to4, bwt_words, sa, sa_intv=32):
        self = cls()
        bwt = np.ascontiguousarray(bwt_words, dtype=np.uint32)
        sa = np.ascontiguousarray(sa, dtype=np.uint64)
        self._keep = [bwt, sa]
        v = self.view
        v.primary = int(primary)
        v.L2[0] = 0
        for i in range(4):
            v.L2[i + 1] = int(L2_1to4[i])
        v.seq_len = int(L2_1to4[3])
        v.bwt_size = bwt.size
        v.bwt = bwt.ctypes.data
        v.sa_intv = sa_intv
        v.n_sa = sa.size
        v.sa = sa.ctypes.data
        return self

    @classmethod
    def build(cls, fwd_nt4, device=0, force_64bit=False, verbose=False):
        """Build the FM-index of a genome (forward strand, codes 0..3) on the GPU (cs_index_build_flags)."""
        self = cls()
        L = load_library()
        g = np.ascontiguousarray(fwd_nt4, dtype=np.uint8)
        h = C.c_void_p()
        _check(L.cs_index_build_flags(g.ctypes.data, g.size, int(device), (1 if force_64bit else 0) | (2 if verbose else 0), C.byref(h)))
        self._handle = h
        _check(L.cs_index_view(h, C.byref(self.view)))
        return self

    def save(self, prefix):
        """Write <prefix>.bwt / <prefix>.sa in the reference's formats (cs_index_save)."""
        if self._handle is None:
            raise ValueError("only file- or GPU-built indexes can be saved")
        _check(load_library().cs_index_save(self._handle, os.fsencode(prefix)))

    def arrays(self):
        """numpy views (copies) of the bwt words and the sampled SA"""
        v = self.view
        return _view(v.bwt, "<u4", int(v.bwt_size)), _view(v.sa, "<u8", int(v.n_sa))

    def close(self):
        if self._handle is not None:
            load_library().cs_index_free(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RefSeq:
    """reference sequences from a FASTA (cs_refseq_t): contig table, holes, forward strand as codes 0..3"""

    def __init__(self, fasta):
        self._h = C.c_void_p()
        _check(load_library().cs_refseq_from_fasta(os.fsencode(fasta), C.byref(self._h)))
        p, n, ns, nh = C.c_void_p(), C.c_uint64(), C.c_int32(), C.c_int32()
        _check(load_library().cs_refseq_codes(self._h, C.byref(p), C.byref(n), C.byref(ns), C.byref(nh)))
        self.l_pac, self.n_seqs, self.n_holes = int(n.value), int(ns.value), int(nh.value)
        self.codes = _view(p.value, np.uint8, self.l_pac)

    def save(self, prefix):
        """<prefix>.pac / .ann / .amb as bwaidx writes them"""
        _check(load_library().cs_refseq_save(self._h, os.fsencode(prefix)))

    def close(self):
        if self._h:
            load_library().cs_refseq_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Reader:
    """chunks of a reordered-reads or FASTQ file (plain or gzip) as the engine takes them (cs_reader_t); iterate for (bases, offsets)
    views of the reader's two alternating buffers: a chunk is valid until the one after the next is read"""

    def __init__(self, path, chunk_bases=10_000_000):
        self._h = C.c_void_p()
        _check(load_library().cs_reader_open(os.fsencode(path), int(chunk_bases), C.byref(self._h)))

    def __iter__(self):
        return self

    def __next__(self):
        b, o, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        _check(load_library().cs_reader_next(self._h, C.byref(b), C.byref(o), C.byref(n)))
        if n.value == 0:
            raise StopIteration
        off = _view(o.value, "<u8", n.value + 1, False)
        return _view(b.value, np.uint8, int(off[-1]), False), off

    def close(self):
        if self._h:
            load_library().cs_reader_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Chainer:
    """mem_chain over a batch's seeds (cs_chainer_t); needs <prefix>.ann"""

    def __init__(self, prefix):
        self._h = C.c_void_p()
        _check(load_library().cs_chainer_create(os.fsencode(prefix), C.byref(self._h)))

    def chain(self, mem_off, mems, seed_off, seeds, read_offsets, params=None, threads=4, copy=True):
        """host CSR seeds (numpy arrays as Result holds them) -> dict(chain_off, chains (CHAIN_DT), cseed_off, cseeds (SEED_DT)), copies"""
        params = params or ChainParams()
        mem_off = np.ascontiguousarray(mem_off, dtype=np.uint64); seed_off = np.ascontiguousarray(seed_off, dtype=np.uint64)
        mems = np.ascontiguousarray(mems, dtype=INTV_DT); seeds = np.ascontiguousarray(seeds, dtype=SEED_DT)
        ro = np.ascontiguousarray(read_offsets, dtype=np.uint64)
        res = CResult(mem_off.size - 1, mems.size, seeds.size, mem_off.ctypes.data, mems.ctypes.data, seed_off.ctypes.data, seeds.ctypes.data)
        out = CChainResult()
        _check(load_library().cs_chain_batch(self._h, C.byref(params), C.byref(res), ro.ctypes.data, int(threads), C.byref(out)))
        return dict(chain_off=_view(out.chain_off, "<u8", int(out.n_reads) + 1, copy), chains=_view(out.chains, CHAIN_DT, int(out.n_chains), copy),
                    cseed_off=_view(out.cseed_off, "<u8", int(out.n_chains) + 1, copy), cseeds=_view(out.cseeds, SEED_DT, int(out.n_seeds), copy))   # copy=False: views of the chainer's arrays, valid until its next chain()

    def filter(self, chain_off, chains, cseed_off, cseeds, bases, read_offsets, params=None, threads=4, copy=True):
        """cs_chain_filter (mem_chain_flt + mem_flt_chained_seeds): chains as chain() returns them -> the same dict for the surviving
        chains in the reference's order, plus cseed_score (int32 per surviving seed); copies"""
        params = params or FltParams()
        chain_off = np.ascontiguousarray(chain_off, dtype=np.uint64); cseed_off = np.ascontiguousarray(cseed_off, dtype=np.uint64)
        chains = np.ascontiguousarray(chains, dtype=CHAIN_DT); cseeds = np.ascontiguousarray(cseeds, dtype=SEED_DT)
        ro = np.ascontiguousarray(read_offsets, dtype=np.uint64)
        bases = None if bases is None else np.ascontiguousarray(bases, dtype=np.uint8)
        cin = CChainResult(chain_off.size - 1, chains.size, cseeds.size, chain_off.ctypes.data, chains.ctypes.data if chains.size else None, cseed_off.ctypes.data,
                           cseeds.ctypes.data if cseeds.size else None)
        out = CChainResult(); sc = C.c_void_p()
        _check(load_library().cs_chain_filter(self._h, C.byref(params), C.byref(cin), bases.ctypes.data if bases is not None and bases.size else None, ro.ctypes.data,
                                              int(threads), C.byref(out), C.byref(sc)))
        return dict(chain_off=_view(out.chain_off, "<u8", int(out.n_reads) + 1, copy), chains=_view(out.chains, CHAIN_DT, int(out.n_chains), copy),
                    cseed_off=_view(out.cseed_off, "<u8", int(out.n_chains) + 1, copy), cseeds=_view(out.cseeds, SEED_DT, int(out.n_seeds), copy),
                    cseed_score=_view(sc.value, "<i4", int(out.n_seeds), copy))

    def close(self):
        if self._h:
            load_library().cs_chainer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_index_from_fasta(fasta, prefix, device=0):
    """bwa_idx_build: all five index files from a FASTA, the suffix sort on the GPU (cs_index_build_fasta)"""
    _check(load_library().cs_index_build_fasta(os.fsencode(fasta), os.fsencode(prefix), int(device)))


class Result:
    """CSR result of one batch as numpy arrays (host variant) or raw device pointers (device variant)."""

    def __init__(self, cres, on_device, want_sal, copy=True):
        self.n_reads, self.n_mems, self.n_seeds = int(cres.n_reads), int(cres.n_mems), int(cres.n_seeds)
        self.on_device = on_device
        self.ptr = dict(mem_off=cres.mem_off, mems=cres.mems, seed_off=cres.seed_off, seeds=cres.seeds)
        if not on_device:
            self.mem_off = _view(cres.mem_off, "<u8", self.n_reads + 1, copy)
            self.mems = _view(cres.mems, INTV_DT, self.n_mems, copy)
            self.seed_off = _view(cres.seed_off, "<u8", self.n_reads + 1, copy) if want_sal else None
            self.seeds = _view(cres.seeds, SEED_DT, self.n_seeds, copy) if want_sal else None


def _view(ptr, dt, n, copy=True):
    dt = np.dtype(dt)
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    a = np.frombuffer(buf, dtype=dt, count=n)
    return a.copy() if copy else a  # copy=False: a view of the engine's pinned buffer, valid until the next call


class PinnedArray(np.ndarray):
    """uint8 numpy array over pinned host memory from cs_host_alloc (freed with the array)"""
    _ptr = None

    def __del__(self):
        if self._ptr:
            try:
                load_library().cs_host_free(C.c_void_p(self._ptr))
            except Exception:
                pass
            self._ptr = None


def pinned_array(nbytes):
    """uint8 array of `nbytes` in pinned host memory (cs_host_alloc): what an integration would read its chunk of reads into"""
    p = C.c_void_p()
    _check(load_library().cs_host_alloc(int(nbytes), C.byref(p)))
    buf = (C.c_uint8 * int(nbytes)).from_address(p.value)
    a = np.frombuffer(buf, dtype=np.uint8).view(PinnedArray)
    a._ptr = p.value
    return a


def pack_reads(bases, offsets, threads=1, scalar=False):
    """cs_pack_reads: the reads as the kernels' 16-byte records of 32 bases -> uint32 array (n_records, 4)"""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    rec = np.zeros(((int(offsets[-1]) >> 5) + n, 4), dtype=np.uint32)
    _check(load_library().cs_pack_reads(bases.ctypes.data, offsets.ctypes.data, n, rec.ctypes.data, int(threads), 1 if scalar else 0))
    return rec


class Engine:
    """One GPU, one resident index (cs_engine_t)."""

    def __init__(self, index, device=0, **options):
        """options: fields of cs_engine_options_t (full_sa, sa64, text_mode, text_arrays, jump_k, kmer_filter, fused, mem_cap,
        lep_arena_mb, max_raw_mb, r3_text_iter, count_sal_merged, verbose)"""
        self._L = load_library()
        self._h = C.c_void_p()
        self._index = index  # keep host arrays alive during upload
        self.options = EngineOptions(**options)
        _check(self._L.cs_engine_create_opts(C.byref(index.view), int(device), C.byref(self.options), C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            self._L.cs_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- hot path
    def seed_batch(self, bases, offsets, params=None, copy=True):
        """Host buffers in, numpy CSR out (cs_engine_seed_batch).  copy=False returns views of the engine's pinned
        result buffers (valid until the next call on this engine) instead of copies."""
        params = params or Params()
        if not isinstance(bases, PinnedArray):
            bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        res = CResult()
        _check(self._L.cs_engine_seed_batch(self._h, C.byref(params), offsets.size - 1, bases.ctypes.data, offsets.ctypes.data, C.byref(res)))
        return Result(res, False, bool(params.want_sal), copy)

    def seed_batch_packed(self, bases, offsets, params=None):
        """Host buffers in, the PACKED CSR (what crosses PCIe) out as views of the engine's pinned buffers: dict(mem_format, mem_off,
        mems (MEM16_DT or INTV_DT), seed_off, seed_rbeg, max_occ); valid until the next call (cs_engine_seed_batch_packed)"""
        params = params or Params()
        if not isinstance(bases, PinnedArray):
            bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        res = CPacked()
        _check(self._L.cs_engine_seed_batch_packed(self._h, C.byref(params), offsets.size - 1, bases.ctypes.data, offsets.ctypes.data, C.byref(res)))
        return self._packed(res)

    def submit(self, bases, offsets, params=None):
        """queue a batch (cs_engine_submit); bases / offsets must be contiguous uint8 / uint64 arrays that stay alive and unchanged until collected"""
        params = params or Params()
        assert bases.dtype == np.uint8 and offsets.dtype == np.uint64 and bases.flags.c_contiguous and offsets.flags.c_contiguous
        self._inflight = getattr(self, "_inflight", []) + [(bases, offsets, params)]
        _check(self._L.cs_engine_submit(self._h, C.byref(params), offsets.size - 1, bases.ctypes.data, offsets.ctypes.data))

    def collect_packed(self):
        """block until the oldest submitted batch is complete (cs_engine_collect_packed); same dict as seed_batch_packed"""
        res = CPacked()
        _check(self._L.cs_engine_collect_packed(self._h, C.byref(res)))
        self._inflight = getattr(self, "_inflight", [])[1:]
        return self._packed(res)

    @staticmethod
    def _packed(res):
        n = int(res.n_reads)
        sal = bool(res.seed_off)
        return PackedResult(n_reads=n, n_mems=int(res.n_mems), n_seeds=int(res.n_seeds), mem_format=int(res.mem_format), max_occ=int(res.max_occ),
                            mem_off=_view(res.mem_off, "<u8", n + 1, False),
                            mems=_view(res.mems, MEM16_DT if res.mem_format == 1 else INTV_DT, int(res.n_mems), False),
                            seed_off=_view(res.seed_off, "<u8", n + 1, False) if sal else None, seed_format=int(res.seed_format),
                            seed_rbeg_lo=_view(res.seed_rbeg_lo, "<u4", int(res.n_seeds), False) if sal else None,
                            seed_rbeg_hi=_view(res.seed_rbeg_hi, "u1", int(res.n_seeds), False) if sal else None)

    def seed_batch_device(self, d_bases, d_offsets, n_reads, n_bases, params=None):
        """Device pointers (ints) in, device pointers out (cs_engine_seed_batch_device): no PCIe traffic in the call."""
        params = params or Params()
        res = CResult()
        _check(self._L.cs_engine_seed_batch_device(self._h, C.byref(params), int(n_reads), C.c_void_p(d_bases), C.c_void_p(d_offsets),
                                                   int(n_bases), C.byref(res)))
        return Result(res, True, bool(params.want_sal))

    def submit_device(self, d_bases, d_offsets, n_reads, n_bases, params=None):
        """cs_engine_submit_device: queue a device-resident batch (up to two in flight, seeded on alternating pass contexts)"""
        params = params or Params()
        self._dev_sal = getattr(self, "_dev_sal", [])
        _check(self._L.cs_engine_submit_device(self._h, C.byref(params), int(n_reads), C.c_void_p(d_bases), C.c_void_p(d_offsets), int(n_bases)))
        self._dev_sal.append(bool(params.want_sal))

    def collect_device(self):
        """cs_engine_collect_device: the oldest submitted device batch, as device pointers (valid until the second submit from now)"""
        res = CResult()
        _check(self._L.cs_engine_collect_device(self._h, C.byref(res)))
        return Result(res, True, self._dev_sal.pop(0))

    def result_digest(self):
        """(mem_off, mems, seed_off, seeds) digests of the result of the last seed call, computed on the device"""
        d = Digest()
        _check(self._L.cs_engine_result_digest(self._h, C.byref(d)))
        return d.astuple()

    def gather_reads(self, read_ids, copy=True):
        """CSR slice of the selected reads of the last result (cs_engine_gather_reads) as a host Result"""
        ids = np.ascontiguousarray(read_ids, dtype=np.uint64)
        res = CResult()
        _check(self._L.cs_engine_gather_reads(self._h, ids.size, ids.ctypes.data, C.byref(res)))
        return Result(res, False, bool(res.seed_off), copy)

    def traffic_model(self):
        """byte model of the SMEM stage since the last reset_stats(): {"kernels": {name: {"bytes": B, "events": {event: n}}}, "stream_bytes": S,
        "bytes": total} (cs_engine_traffic_model)"""
        t = Traffic()
        _check(self._L.cs_engine_traffic_model(self._h, C.byref(t)))
        eb = [int(x) for x in t.event_bytes]
        out = {"kernels": {}, "stream_bytes": int(t.stream_bytes), "event_bytes": dict(zip(EVENTS, eb))}
        tot = int(t.stream_bytes)
        for k, name in enumerate(KERNELS):
            ev = [int(x) for x in t.events[k]]
            if any(ev):
                b = sum(n * w for n, w in zip(ev, eb))
                out["kernels"][name] = {"bytes": b, "events": {e: n for e, n in zip(EVENTS, ev) if n}}
                tot += b
        out["bytes"] = tot
        return out

    def check_index(self, d_fwd_nt4=None, l_pac=0):
        """cs_engine_check_index: violation counts of the resident index against the text (and, when given, the caller's genome on the device)"""
        c = IndexCheck()
        _check(self._L.cs_engine_check_index(self._h, int(d_fwd_nt4) if d_fwd_nt4 else None, int(l_pac), C.byref(c)))
        return {n: int(getattr(c, n)) for n, _ in IndexCheck._fields_ if n != "reserved"}

    def stats(self):
        st = Stats()
        _check(self._L.cs_engine_stats(self._h, C.byref(st)))
        return st.asdict()

    def reset_stats(self):
        self._L.cs_engine_reset_stats(self._h)

    # ---- primitives (parity tests of the building blocks)
    def occ4(self, k):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        out = np.zeros((k.size, 4), dtype=np.uint64)
        _check(self._L.cs_engine_occ4(self._h, k.size, k.ctypes.data, out.ctypes.data))
        return out

    def extend(self, ik, is_back):
        ik = np.ascontiguousarray(ik, dtype=INTV_DT)
        fb = np.ascontiguousarray(is_back, dtype=np.uint8)
        out = np.zeros((ik.size, 4), dtype=INTV_DT)
        _check(self._L.cs_engine_extend(self._h, ik.size, ik.ctypes.data, fb.ctypes.data, out.ctypes.data))
        return out

    def sa(self, k):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        out = np.zeros(k.size, dtype=np.uint64)
        _check(self._L.cs_engine_sa(self._h, k.size, k.ctypes.data, out.ctypes.data))
        return out

    def probe_random_lines(self, waves_per_simd=8, steps=2000):
        """64-byte random Occ-block reads per second in dependent chains (cs_engine_probe_random_lines)"""
        out = C.c_double()
        _check(self._L.cs_engine_probe_random_lines(self._h, waves_per_simd, steps, C.byref(out)))
        return out.value

    # ---- device memory helpers
    def alloc(self, nbytes):
        p = C.c_void_p()
        _check(self._L.cs_device_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def free(self, dptr):
        _check(self._L.cs_device_free(self._h, C.c_void_p(dptr)))

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        _check(self._L.cs_device_upload(self._h, C.c_void_p(dptr), arr.ctypes.data, arr.nbytes))

    def download(self, dptr, dtype, count):
        out = np.zeros(count, dtype=dtype)
        _check(self._L.cs_device_download(self._h, out.ctypes.data, C.c_void_p(dptr), out.nbytes))
        return out

    def sync(self):
        _check(self._L.cs_device_sync(self._h))


class Extender:
    """cs_extender_t: the banded Smith-Waterman seed extension on the GPU (include/compseed_amd.h; the reference's BandedPairWiseSW)"""

    def __init__(self, device=0, params=None):
        self.L = load_library()
        self.h = C.c_void_p()
        _check(self.L.cs_extender_create(device, C.byref(params) if params is not None else None, C.byref(self.h)))

    def extend(self, pairs, qbuf, tbuf, w=100):
        """pairs: EXT_PAIR_DT array; qbuf / tbuf: uint8 codes 0..4.  Returns an EXT_RES_DT array (raises CSError on bad pairs)."""
        pairs = np.ascontiguousarray(pairs, dtype=EXT_PAIR_DT)
        qbuf = np.ascontiguousarray(qbuf, dtype=np.uint8); tbuf = np.ascontiguousarray(tbuf, dtype=np.uint8)
        out = np.zeros(pairs.size, dtype=EXT_RES_DT)
        self.last_rc = self.L.cs_extend_batch(self.h, pairs.size, pairs.ctypes.data if pairs.size else None, qbuf.ctypes.data if qbuf.size else None, qbuf.size,
                                              tbuf.ctypes.data if tbuf.size else None, tbuf.size, int(w), out.ctypes.data if pairs.size else None)
        self.last_out = out
        _check(self.last_rc)
        return out

    def extend_device(self, d_pairs, n, d_q, q_bytes, d_t, t_bytes, d_out, w=100):
        _check(self.L.cs_extend_batch_device(self.h, int(n), int(d_pairs), int(d_q), int(q_bytes), int(d_t), int(t_bytes), int(w), int(d_out)))

    def stats(self):
        st = ExtStats()
        _check(self.L.cs_extender_stats(self.h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in ExtStats._fields_}

    def close(self):
        if self.h:
            self.L.cs_extender_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class Aligner:
    """cs_aligner_t: the extension stage as a whole (the reference's mem_chain2aln_across_reads_V2): chains in, alignment regions out"""

    def __init__(self, prefix, device=0, params=None):
        self.L = load_library()
        self.h = C.c_void_p()
        _check(self.L.cs_aligner_create(os.fsencode(prefix), device, C.byref(params) if params is not None else None, C.byref(self.h)))

    def extend_chains(self, chain_off, chains, cseed_off, cseeds, bases, read_offsets, cseed_score=None, copy=True):
        """chains as Chainer.chain returns them (or the caller's own, e.g. filtered); returns dict(reg_off, regs (ALNREG_DT)), copies"""
        chain_off = np.ascontiguousarray(chain_off, dtype=np.uint64); cseed_off = np.ascontiguousarray(cseed_off, dtype=np.uint64)
        chains = np.ascontiguousarray(chains, dtype=CHAIN_DT); cseeds = np.ascontiguousarray(cseeds, dtype=SEED_DT)
        bases = np.ascontiguousarray(bases, dtype=np.uint8); ro = np.ascontiguousarray(read_offsets, dtype=np.uint64)
        sc = None if cseed_score is None else np.ascontiguousarray(cseed_score, dtype=np.int32)
        cr = CChainResult(chain_off.size - 1, chains.size, cseeds.size, chain_off.ctypes.data, chains.ctypes.data if chains.size else None, cseed_off.ctypes.data,
                          cseeds.ctypes.data if cseeds.size else None)
        out = CAlnResult()
        _check(self.L.cs_extend_chains(self.h, C.byref(cr), sc.ctypes.data if sc is not None and sc.size else None, bases.ctypes.data if bases.size else None, ro.ctypes.data, C.byref(out)))
        return dict(reg_off=_view(out.reg_off, "<u8", int(out.n_reads) + 1, copy), regs=_view(out.regs, ALNREG_DT, int(out.n_regs), copy))

    def dedup_regions(self, reg_off, regs, bases, read_offsets, params=None, copy=True):
        """cs_dedup_regions (purged regions dropped, mem_sort_dedup_patch): regions as extend_chains returns them -> dict(reg_off, regs, n_comp), copies"""
        params = params or DedupParams()
        reg_off = np.ascontiguousarray(reg_off, dtype=np.uint64); regs = np.ascontiguousarray(regs, dtype=ALNREG_DT)
        bases = np.ascontiguousarray(bases, dtype=np.uint8); ro = np.ascontiguousarray(read_offsets, dtype=np.uint64)
        cin = CAlnResult(reg_off.size - 1, regs.size, reg_off.ctypes.data, regs.ctypes.data if regs.size else None)
        out = CAlnResult(); nc = C.c_void_p()
        _check(self.L.cs_dedup_regions(self.h, C.byref(params), C.byref(cin), bases.ctypes.data if bases.size else None, ro.ctypes.data, C.byref(out), C.byref(nc)))
        return dict(reg_off=_view(out.reg_off, "<u8", int(out.n_reads) + 1, copy), regs=_view(out.regs, ALNREG_DT, int(out.n_regs), copy), n_comp=_view(nc.value, "<i4", int(out.n_regs), copy))

    def stats(self):
        st = AlnStats()
        _check(self.L.cs_aligner_stats(self.h, C.byref(st)))
        return {n: (float(getattr(st, n)) if n == "ext_kernel_ms" else int(getattr(st, n))) for n, _ in AlnStats._fields_}

    def close(self):
        if self.h:
            self.L.cs_aligner_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
