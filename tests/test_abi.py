"""CPU-side checks of the drop-in boundary: the library loads and exports exactly the declared C ABI."""
import os
import re

import pytest

import _data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import compseed_amd as ca
    if not os.path.exists(ca.lib_path()):
        ca.build_library()
    return ca.load_library()


def test_header_symbols_are_exported(lib):
    import compseed_amd.binding as b
    hdr = open(os.path.join(ROOT, "include", "compseed_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(b.SYMBOLS)
    inline = set(re.findall(r"static inline [a-z0-9_ ]+?\b(cs_[a-z0-9_]+)\s*\(", hdr))   # helpers that live in the header itself
    assert inline == {"cs_unpack_mem", "cs_mem_seed_count", "cs_packed_seed_rbeg"}
    for name in declared:
        if name not in inline:
            assert hasattr(lib, name), name


def test_index_loader_reads_reference_formats(lib):
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    f = _data.load_bwt_files()
    assert ix.view.primary == f["primary"] and ix.view.seq_len == f["seq_len"]
    assert list(ix.view.L2) == [0] + [int(x) for x in f["L2"]]
    assert ix.view.bwt_size == f["bwt"].size and ix.view.n_sa == f["sa"].size and ix.view.sa_intv == 32
    ix.close()


def test_index_loader_errors(lib, tmp_path):
    import compseed_amd as ca
    with pytest.raises(ca.CSError) as ei:
        ca.Index.load(str(tmp_path / "nope"))
    assert ei.value.code == -2
    # SA that belongs to another BWT: "SA-BWT inconsistency" (bwt.c:429)
    import shutil
    shutil.copy(_data.PREFIX + ".bwt", tmp_path / "x.bwt")
    raw = bytearray(open(_data.PREFIX + ".sa", "rb").read())
    raw[0] ^= 1
    open(tmp_path / "x.sa", "wb").write(raw)
    with pytest.raises(ca.CSError) as ei:
        ca.Index.load(str(tmp_path / "x"))
    assert "inconsistency" in str(ei.value)


def test_no_gpu_fails_loudly(lib):
    """The product has no CPU path: on a box without a GPU engine creation must raise, never fall back."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    import compseed_amd as ca
    ix = ca.Index.load(_data.PREFIX)
    with pytest.raises(ca.CSError) as ei:
        ca.Engine(ix, 0)
    assert ei.value.code == -4
    ix.close()


def test_params_default(lib):
    import compseed_amd as ca
    p = ca.Params(k=1, r=9, s=9, c=9, y=9)
    lib.cs_params_default(p)
    assert (p.min_seed_len, p.split_width, p.max_occ, p.max_mem_intv, p.want_sal) == (19, 10, 500, 20, 1)
    assert abs(p.split_factor - 1.5) < 1e-9


def test_engine_options_default(lib):
    import compseed_amd as ca
    o = ca.EngineOptions()
    assert (o.full_sa, o.sa64, o.text_mode, o.text_arrays, o.jump_k, o.kmer_filter, o.fused, o.mem_cap) == (1, 0, 1, 1, 15, 1, 0, 64)
    assert (o.lep_arena_mb, o.max_raw_mb, o.count_sal_merged, o.verbose) == (16384, 24576, 0, 0) and not any(o.reserved)
    assert (o.pipeline_reads, o.expand_threads, o.host_pack_threads, o.passes_in_flight) == (5000000, 16, 8, 2)
    with pytest.raises(TypeError):
        ca.EngineOptions(no_such_option=1)


def test_ctypes_structs_match_the_header(lib, tmp_path):
    """sizeof / offsetof of every struct of include/compseed_amd.h as gcc sees them == the ctypes mirrors in binding.py"""
    import ctypes
    import subprocess
    import compseed_amd.binding as b
    structs = {"cs_index_view_t": b.IndexView, "cs_params_t": b.Params, "cs_result_t": b.CResult, "cs_stats_t": b.Stats,
               "cs_engine_options_t": b.EngineOptions, "cs_traffic_t": b.Traffic, "cs_digest_t": b.Digest, "cs_packed_result_t": b.CPacked, "cs_ext_params_t": b.ExtParams, "cs_ext_stats_t": b.ExtStats,
               "cs_index_check_t": b.IndexCheck, "cs_flt_params_t": b.FltParams, "cs_dedup_params_t": b.DedupParams, "cs_aln_params_t": b.AlnParams, "cs_aln_result_t": b.CAlnResult, "cs_aln_stats_t": b.AlnStats}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "compseed_amd.h"', 'int main(void) {']
    for cname, st in structs.items():
        src.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in st._fields_:
            src.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src.append("return 0; }")
    c = tmp_path / "abi.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(c)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, st in structs.items():
        assert int(got[cname]) == ctypes.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(st, fname).offset, (cname, fname)
    # the two array-element structs of the extension entry points are numpy dtypes on the Python side
    for cname, dt in (("cs_ext_pair_t", b.EXT_PAIR_DT), ("cs_ext_result_t", b.EXT_RES_DT), ("cs_alnreg_t", b.ALNREG_DT)):
        src2 = ['#include <stdio.h>', '#include <stddef.h>', '#include "compseed_amd.h"', 'int main(void) {', 'printf("%%zu\\n", sizeof(%s));' % cname]
        src2 += ['printf("%%zu\\n", offsetof(%s, %s));' % (cname, f) for f in dt.names] + ["return 0; }"]
        c2 = tmp_path / (cname + ".c"); c2.write_text("\n".join(src2))
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(tmp_path / cname), str(c2)], check=True)
        nums = [int(x) for x in subprocess.run([str(tmp_path / cname)], capture_output=True, text=True, check=True).stdout.split()]
        assert nums[0] == dt.itemsize and nums[1:] == [dt.fields[f][1] for f in dt.names], cname


def test_product_reads_no_environment_switches():
    """configuration goes through cs_params_t / cs_engine_options_t, not through getenv"""
    for base, _, files in os.walk(os.path.join(ROOT, "compseed_amd", "csrc")):
        for f in files:
            if f.endswith((".hip", ".hpp", ".cpp", ".h", ".c")):
                assert "getenv" not in open(os.path.join(base, f), errors="replace").read(), f


def test_product_does_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing under compseed_amd/ may include, link or load it."""
    for base, _, files in os.walk(os.path.join(ROOT, "compseed_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(base, f), errors="replace").read()
                assert "cs_oracle" not in txt and "libcsoracle" not in txt and "oracle/" not in txt, os.path.join(base, f)
