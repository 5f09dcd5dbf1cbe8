#!/usr/bin/env bash
# Build tests/hostsim/libfastp_gpu_sim.so: the PRODUCT sources compiled by g++ against the
# fake HIP runtime in this directory (TEST INFRASTRUCTURE ONLY - see hip/hip_runtime.h).
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
SRC=$HERE/../../fastp_amd/csrc
# one build at a time: pytest-xdist workers (and the build step of several tests) may call this script concurrently
exec 9>"$HERE"/.build.lock
flock 9
g++ -std=c++17 -O2 -g -fPIC -shared -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-unused-variable \
    -I"$HERE" -I"$SRC" -x c++ "$SRC/fastp_gpu.hip" "$SRC/fq_host.cpp" "$SRC/fq_glue.cpp" "$SRC/fq_comm.cpp" "$SRC/fq_stream.cpp" "$HERE/sim.cpp" \
    -ldl -lpthread -lz -o "$HERE/libfastp_gpu_sim.so"
# the in-process stand-in for librccl the collectives test loads through FASTP_GPU_RCCL_LIB
g++ -std=c++17 -O2 -fPIC -shared -Wall "$HERE/../rccl_stub/rccl_stub.cpp" -o "$HERE/../rccl_stub/librccl_stub.so"
echo "built $HERE/libfastp_gpu_sim.so"
