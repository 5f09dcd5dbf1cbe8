#!/usr/bin/env bash
# Build oracle/_ref/fastp_ref_gpu: the REAL reference (OpenGene/fastp v1.3.6) with its two worker-loop bodies
# bound to libfastp_gpu.so (TEST INFRASTRUCTURE - the drop-in boundary exercised end to end).
#
# Every reference source is compiled where it lies under /root/reference/src, except peprocessor.cpp,
# seprocessor.cpp and evaluator.cpp, of which patched copies are generated into oracle/_ref/src_gpu/ (git-ignored) by
# oracle/patches/apply_gpu_worker.py: four inserted lines that call oracle/patches/gpu_worker.cpp.
# Same shims as build_ref.sh (scalar simd, ISA-L inflate over zlib).  Needs fastp_amd/libfastp_gpu.so (__graft_entry__.build()).
set -euo pipefail
REF=${FASTP_REFERENCE_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/.." && pwd)
OUT=$HERE/_ref
OBJ=$OUT/obj_gpu
GEN=$OUT/src_gpu
if [ ! -d "$REF/src" ]; then
    echo "build_ref_gpu: $REF/src not present (GPU box?) - keeping prebuilt $OUT/fastp_ref_gpu" >&2
    exit 0
fi
mkdir -p "$OBJ" "$GEN"
mkdir -p "$OUT"
# one build at a time: pytest-xdist workers (and the build step of several tests) may call this script concurrently
exec 9>"$OUT"/.build.lock
flock 9
python3 "$HERE/patches/apply_gpu_worker.py" "$REF" "$GEN" > /dev/null
CXX=${CXX:-g++}
CXXFLAGS="-std=c++11 -pthread -O3 -w -I$REF -I$REF/src -I$HERE/shims -I/opt/conda/include -I$HERE/patches -I$ROOT/include"
pids=()
for f in "$REF"/src/*.cpp; do
    b=$(basename "$f" .cpp)
    [ "$b" = simd ] && continue
    src=$f
    { [ "$b" = peprocessor ] || [ "$b" = seprocessor ] || [ "$b" = evaluator ] || [ "$b" = fastqreader ] || [ "$b" = duplicate ]; } && src=$GEN/$b.cpp
    o=$OBJ/$b.o
    if [ ! -f "$o" ] || [ "$src" -nt "$o" ] || [ "$HERE/patches/gpu_worker.h" -nt "$o" ] || [ "$HERE/shims/isa-l/igzip_lib.h" -nt "$o" ]; then
        $CXX $CXXFLAGS -c "$src" -o "$o" &
        pids+=($!)
    fi
done
$CXX $CXXFLAGS -c "$HERE/shims/simd_scalar.cpp" -o "$OBJ/simd_scalar.o" &
pids+=($!)
$CXX $CXXFLAGS -c "$HERE/patches/gpu_worker.cpp" -o "$OBJ/gpu_worker.o" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
# the binary finds the engine next to the repo's package whatever directory it is started from
# libdeflate: the system's runtime library by path (conda's lib directory would bring its older libstdc++ along,
# which the HIP runtime behind libfastp_gpu.so cannot live with)
DEFLATE=/usr/lib/x86_64-linux-gnu/libdeflate.so.0
[ -f "$DEFLATE" ] || DEFLATE=/opt/conda/lib/libdeflate.so
# (linked beside their names, then renamed: a test process may be executing the old binaries at this moment)
$CXX -pthread "$OBJ"/*.o -o "$OUT/fastp_ref_gpu.tmp.$$" "$DEFLATE" -lz -lpthread \
    -L"$ROOT/fastp_amd" -Wl,-rpath,'$ORIGIN/../../fastp_amd' -lfastp_gpu
mv -f "$OUT/fastp_ref_gpu.tmp.$$" "$OUT/fastp_ref_gpu"
echo "built $OUT/fastp_ref_gpu"
# the same binding against the SIMT emulator build of the engine (tests/hostsim): lets the CPU-only suite run
# the patched reference end to end on small inputs
SIM=$ROOT/tests/hostsim/libfastp_gpu_sim.so
if [ -f "$SIM" ]; then
    $CXX -pthread "$OBJ"/*.o -o "$OUT/fastp_ref_gpusim.tmp.$$" "$DEFLATE" -lz -lpthread "$SIM" -Wl,-rpath,"$ROOT/tests/hostsim"
    mv -f "$OUT/fastp_ref_gpusim.tmp.$$" "$OUT/fastp_ref_gpusim"
    echo "built $OUT/fastp_ref_gpusim"
fi
