#!/bin/bash
# An A/B build of the library next to the product one: tools/build_ab.sh NAME -DFLAG=VALUE ...  -> fastp_amd/libfastp_gpu_NAME.so
# (compile-time variants that an environment switch cannot select; a visit script then runs both through FASTP_GPU_LIB)
set -e
cd "$(dirname "$0")/.."
C=fastp_amd/csrc
NAME=$1; shift
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wall -Wno-unused-function "$@" \
  $C/fastp_gpu.hip $C/fq_host.cpp $C/fq_glue.cpp $C/fq_comm.cpp $C/fq_stream.cpp -ldl -lpthread -lz -o fastp_amd/libfastp_gpu_$NAME.so
echo built fastp_amd/libfastp_gpu_$NAME.so
