"""compseed_amd -- MI355X-native compressive SMEM seeding (host-side Python mirror of include/compseed_amd.h).

The product is the HIP library `libcompseed_amd.so` behind a plain C ABI; this package only binds it with ctypes
so that tests, bench.py and Python callers can drive the same entry points a C/C++ host (CompSeed's
seed_and_extend) would.  There is no CPU fallback: importing works anywhere, but creating an
Engine without the built library or without a GPU raises.
"""
from . import binding  # noqa: F401
from .binding import (CSError, Engine, EngineOptions, Index, Params, Result, Stats, lib_path, load_library, build_library,  # noqa: F401
                      disable_mask, pinned_array, pack_reads, RefSeq, Reader, Chainer, ChainParams, FltParams, DedupParams, build_index_from_fasta, unpack_mems16, INTV_DT, SEED_DT, MEM16_DT,
                      Extender, ExtParams, EXT_PAIR_DT, EXT_RES_DT, packed_rbeg, Aligner, AlnParams, ALNREG_DT, CHAIN_DT)

__all__ = ["CSError", "Engine", "EngineOptions", "Index", "Params", "Result", "Stats", "lib_path", "load_library", "build_library",
           "disable_mask", "pinned_array", "pack_reads", "RefSeq", "Reader", "Chainer", "ChainParams", "FltParams", "DedupParams", "build_index_from_fasta", "unpack_mems16", "INTV_DT", "SEED_DT", "MEM16_DT",
           "Extender", "ExtParams", "EXT_PAIR_DT", "EXT_RES_DT", "packed_rbeg", "Aligner", "AlnParams", "ALNREG_DT", "CHAIN_DT"]
