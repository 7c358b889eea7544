"""The REAL reference driving the REAL library on the MI355X: integration/_build/CompSeed.gpu is the reference with integration/compseed_gpu.patch
applied (mapping/comp_seed.cpp:2242-2347 take their mems and seeds from cs_engine_submit / cs_engine_collect_packed; main.cpp:60-126 submits each
chunk from the reader step), linked against compseed_amd/libcompseed_amd.so; CompSeed.ref is the unpatched reference.  Both are compiled in the
build container from /root/reference (integration/apply_and_build.sh) and travel to the GPU box as binaries, like oracle/_ref -- no reference
source does.  Byte-identical SAM closes the last link that tests/test_integration.py can only check through a mock: submit from the reader
thread, collect in mem_process_seqs on another thread, two chunks in flight against the asynchronous engine.  The patch also replaces the
reference's extension stage (mem_chain2aln_across_reads_V2, comp_seed.cpp:2371) by cs_extend_chains: the SAM must stay the same with the
alignment regions coming from the library, too -- incl. reads with indels, 800-1500-base and 3-kb reads, and other scoring parameters."""
import os
import subprocess

import pytest

import _data

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "integration", "_build")


def _sam(exe, reads, *flags):
    path = os.path.join(os.path.dirname(_data.GOLD), reads + ".txt") if "/" in reads else os.path.join(_data.GOLD, reads + ".txt")
    r = subprocess.run([exe, "-t", "2", *flags, _data.PREFIX, path], capture_output=True, timeout=900, cwd="/tmp",
                       env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout, r.stderr.decode(errors="replace")


@pytest.mark.parametrize("reads,flags", [("main100", ()), ("sorted150", ()), ("ragged", ()), ("repeat100", ()),
                                         ("main100", ("-K", "20000")), ("sorted150", ("-k", "25", "-r", "1.0", "-y", "5")), ("main100", ("-c", "50")),
                                         ("sorted150", ("-K", "7000", "-t", "4")),
                                         ("aln1/indel150_400", ()), ("flt1/long90", ()), ("ddp1/gap3k", ()), ("aln1/indel150_400", ("-A", "2", "-B", "5", "-O", "7,7", "-w", "30", "-d", "40")),
                                         ("main100", ("-L", "3,3", "-d", "50"))])
def test_patched_reference_on_the_real_engine_writes_the_same_sam(reads, flags):
    ref, gpu = os.path.join(BUILD, "CompSeed.ref"), os.path.join(BUILD, "CompSeed.gpu")
    if not (os.path.exists(ref) and os.path.exists(gpu)):
        pytest.skip("integration/_build/ was not shipped (built by integration/apply_and_build.sh in the build container)")
    want, _ = _sam(ref, reads, *flags)
    got, err = _sam(gpu, reads, *flags)
    assert "GPU seeding:" in err and "no GPU seeding engine" not in err      # the engine branch of the patch did run, on the real library
    assert "GPU extension:" in err and "no GPU extension stage" not in err   # ... and so did the extension stage
    assert got == want and want.count(b"\n") > 100
