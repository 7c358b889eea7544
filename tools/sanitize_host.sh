#!/bin/bash
# tools/sanitize_host.sh -- the host-side sources of the library (everything that is not a kernel file) under AddressSanitizer and
# UndefinedBehaviorSanitizer, run through the CPU test suite (the GPU pool offers no sanitizer runs; the kernels are covered by the
# parity tests).  Builds /tmp/cs_asan/libcompseed_amd_asan.so from the instrumented .cpp objects + the regular kernel objects and runs
# `pytest -m "not gpu"` against it (CS_LIB).  -DCS_FLT_SELFCHECK: cs_chain_filter runs its overlap scan twice, the fast form against the plain loop, and aborts on a difference.  A finding stops the run; the report is in /tmp/cs_asan/{asan,ubsan}.log.*
set -e
R=$(cd "$(dirname "$0")/.." && pwd); S=$R/compseed_amd/csrc; O=/tmp/cs_asan; mkdir -p $O; rm -f $O/*.log.*
make -C $S -j4 all > /dev/null
for f in refseq reader chain chain_filter align dedup host_pack; do
	/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -Wno-option-ignored -DCS_FLT_SELFCHECK -c -o $O/$f.o $S/$f.cpp &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -fsanitize=address,undefined -shared-libasan -Wno-option-ignored -o $O/libcompseed_amd_asan.so $S/build/engine.o $S/build/index_build.o $S/build/extend.o $S/build/align_gpu.o \
	$O/refseq.o $O/reader.o $O/chain.o $O/chain_filter.o $O/align.o $O/dedup.o $O/host_pack.o -lz
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so)
cd $R
CS_LIB=$O/libcompseed_amd_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=$O/asan.log UBSAN_OPTIONS=print_stacktrace=1:log_path=$O/ubsan.log \
	python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
ls $O/*.log.* 2>/dev/null && { echo "sanitizer findings above"; exit 1; } || echo "no sanitizer findings"
