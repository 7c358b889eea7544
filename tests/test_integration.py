"""integration/compseed_gpu.patch as a checked artefact (build container only: needs /root/reference; skipped elsewhere).

integration/apply_and_build.sh applies the patch to a scratch copy of the reference and builds the unpatched reference, the patched
one linked against the REAL libcompseed_amd.so, and the patched one linked against integration/mock_engine.c (the same C ABI
implemented on the oracle).  The mock lets the reference-side binding run end to end without a GPU: the SAM it writes must be
byte-identical to the unpatched reference's, i.e. mems and seeds handed over through cs_engine_submit / cs_engine_collect_packed /
cs_unpack_mem are usable by, and indistinguishable to, the reference's chaining, extension and SAM stages."""
import os
import subprocess

import pytest

import _data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "integration", "_build")

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists only in the build container")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run([os.path.join(ROOT, "integration", "apply_and_build.sh")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return {k: os.path.join(BUILD, "CompSeed." + k) for k in ("ref", "gpu", "mock")}


def _sam(exe, reads, *flags):
    r = subprocess.run([exe, "-t", "2", *flags, _data.PREFIX, os.path.join(_data.GOLD, reads + ".txt")], capture_output=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout, r.stderr.decode(errors="replace")


def test_patch_applies_and_links_against_the_real_library(built):
    out = subprocess.run(["nm", "-D", "--undefined-only", built["gpu"]], capture_output=True, text=True).stdout
    for sym in ("cs_engine_create", "cs_engine_submit", "cs_engine_collect_packed", "cs_engine_destroy", "cs_host_alloc", "cs_last_error"):
        assert sym in out
    assert "libcompseed_amd.so" in subprocess.run(["ldd", built["gpu"]], capture_output=True, text=True).stdout


@pytest.mark.parametrize("reads,flags", [("main100", ()), ("sorted150", ()), ("ragged", ()), ("repeat100", ()),
                                         ("main100", ("-K", "20000")), ("sorted150", ("-k", "25", "-r", "1.0", "-y", "5")), ("main100", ("-c", "50"))])
def test_patched_reference_writes_the_same_sam(built, reads, flags):
    want, _ = _sam(built["ref"], reads, *flags)
    got, err = _sam(built["mock"], reads, *flags)
    assert "GPU seeding:" in err                      # the engine branch of the patch did run (cs_gpu_finish prints its counters)
    assert got == want and want.count(b"\n") > 100


def test_patched_reference_without_a_gpu_keeps_its_own_cpu_path(built):
    """the real library has no CPU path: in this container cs_engine_create fails, the patched reference says so and seeds as before"""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    want, _ = _sam(built["ref"], "sorted150")
    got, err = _sam(built["gpu"], "sorted150")
    assert "no GPU seeding engine" in err and got == want
