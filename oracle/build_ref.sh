#!/usr/bin/env bash
# Build the REAL reference (OpenGene/fastp v1.3.6) into oracle/_ref/fastp_ref.
#
# TEST INFRASTRUCTURE ONLY - never linked or executed by the product path.
#
# The reference sources are compiled where they lie under /root/reference/src
# (nothing is copied into this repo).  Two third-party dependencies of the
# reference are absent from this image, so two tiny shims of our own are used:
#   * oracle/shims/simd_scalar.cpp   replaces src/simd.cpp (Google Highway 1.3.0)
#   * oracle/shims/isa-l/igzip_lib.h provides the ISA-L v2.31.1 inflate calls the reader names, over zlib
# libdeflate comes from /opt/conda (v1.8).  We do NOT run the reference Makefile.
set -euo pipefail
REF=${FASTP_REFERENCE_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
OBJ=$OUT/obj
if [ ! -d "$REF/src" ]; then
    echo "build_ref: $REF/src not present (GPU box?) - keeping prebuilt $OUT/fastp_ref" >&2
    exit 0
fi
mkdir -p "$OBJ"
mkdir -p "$OUT"
# one build at a time: pytest-xdist workers (and the build step of several tests) may call this script concurrently
exec 9>"$OUT"/.build.lock
flock 9
CXX=${CXX:-g++}
CXXFLAGS="-std=c++11 -pthread -O3 -w -I$REF -I$HERE/shims -I/opt/conda/include"
pids=()
for f in "$REF"/src/*.cpp; do
    b=$(basename "$f" .cpp)
    [ "$b" = simd ] && continue
    o=$OBJ/$b.o
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/shims/isa-l/igzip_lib.h" -nt "$o" ]; then
        $CXX $CXXFLAGS -c "$f" -o "$o" &
        pids+=($!)
    fi
done
$CXX $CXXFLAGS -c "$HERE/shims/simd_scalar.cpp" -o "$OBJ/simd_scalar.o" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
# (linked beside its name, then renamed: a test process may be executing the old binary at this moment)
$CXX -pthread "$OBJ"/*.o -o "$OUT/fastp_ref.tmp.$$" -L/opt/conda/lib -Wl,-rpath,/opt/conda/lib -ldeflate -lz -lpthread
mv -f "$OUT/fastp_ref.tmp.$$" "$OUT/fastp_ref"
echo "built $OUT/fastp_ref"
