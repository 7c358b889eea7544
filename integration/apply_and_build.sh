#!/bin/bash
# integration/apply_and_build.sh -- BUILD CONTAINER ONLY (needs /root/reference; the GPU box never runs this).
#
# Applies integration/compseed_gpu.patch to a scratch copy of the reference and builds, file by file with gcc/g++ (the reference's
# own cmake build is not used; -fcommon as in oracle/Makefile):
#   _build/CompSeed.ref    the unpatched reference
#   _build/CompSeed.gpu    the patched reference linked against the real compseed_amd/libcompseed_amd.so (runs where a GPU is)
#   _build/CompSeed.mock   the patched reference linked against integration/mock_engine.c (the same ABI on the oracle, CPU): lets the
#                          patch be exercised end to end here and its SAM compared with CompSeed.ref (tests/test_integration.py)
# Nothing of the reference is copied into the repository: the scratch tree lives under $TMPDIR and is deleted.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd); REPO=$(dirname "$HERE"); REF=${REF:-/root/reference}
OUT=$HERE/_build; mkdir -p "$OUT"
[ -d "$REF" ] || { echo "no reference at $REF" >&2; exit 2; }
make -s -C "$REPO/oracle" oracle
[ -f "$REPO/compseed_amd/libcompseed_amd.so" ] || make -s -C "$REPO/compseed_amd/csrc" all
W=$(mktemp -d); trap 'rm -rf "$W"' EXIT
cp -r "$REF" "$W/ref"; cp -r "$REF" "$W/pat"
( cd "$W/pat" && patch -s -p1 < "$HERE/compseed_gpu.patch" )
C_SRC="FM_index/bwt.c FM_index/bntseq.c FM_index/bwt_gen.c FM_index/is.c FM_index/QSufSort.c FM_index/rle.c FM_index/rope.c bwalib/bwa.c bwalib/utils.c bwalib/ksw.c bwalib/bwashm.c bwalib/kopen.c cstl/kstring.c cstl/kthread.c"
CXX_SRC="mapping/comp_seed.cpp mapping/SST.cpp mapping/bandedSWA.cpp mapping/memcpy_bwamem.cpp main.cpp"
FLAGS="-O2 -g0 -fcommon -mavx2 -w"
build_tree() { # $1 = tree, $2 = extra include
	local t=$1 objs=""
	for f in $C_SRC; do gcc $FLAGS -I"$t" -c -o "$t/${f%.c}.o" "$t/$f" & objs="$objs $t/${f%.c}.o"; done
	for f in $CXX_SRC; do g++ $FLAGS -std=c++11 -I"$t" $2 -c -o "$t/${f%.cpp}.o" "$t/$f" & objs="$objs $t/${f%.cpp}.o"; done
	wait
	echo "$objs"
}
OBJ_REF=$(build_tree "$W/ref" "")
OBJ_PAT=$(build_tree "$W/pat" "-I$REPO/include")
g++ $FLAGS -o "$OUT/CompSeed.ref" $OBJ_REF -lz -lpthread -lm -lrt
# the real library: resolved at run time from the repository (rpath); its own dependencies (libamdhip64) are not needed to link
g++ $FLAGS -o "$OUT/CompSeed.gpu" $OBJ_PAT -L"$REPO/compseed_amd" -lcompseed_amd -Wl,-rpath,"$REPO/compseed_amd" -Wl,--allow-shlib-undefined -lz -lpthread -lm -lrt
# the mock: same header, oracle underneath
gcc -O2 -g0 -std=gnu11 -Wall -I"$REPO/include" -c -o "$W/mock_engine.o" "$HERE/mock_engine.c"
g++ $FLAGS -o "$OUT/CompSeed.mock" $OBJ_PAT "$W/mock_engine.o" -L"$REPO/oracle" -lcsoracle -Wl,-rpath,"$REPO/oracle" -lz -lpthread -lm -lrt
echo "built: $OUT/CompSeed.ref $OUT/CompSeed.gpu $OUT/CompSeed.mock"
