#!/bin/bash
# The PROFILING build of the library: the same sources with -DFQ_PROFILE_ABLATION, in which FASTP_GPU_DEBUG_SKIP leaves steps out of
# the kernels (the measured floors under profiles/; results are meaningless then).  The product library (__graft_entry__.build())
# has no such switch.  Used by the visit scripts through FASTP_GPU_LIB=fastp_amd/libfastp_gpu_abl.so.
set -e
cd "$(dirname "$0")/.."
C=fastp_amd/csrc
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wall -Wno-unused-function -DFQ_PROFILE_ABLATION \
  $C/fastp_gpu.hip $C/fq_host.cpp $C/fq_glue.cpp $C/fq_comm.cpp $C/fq_stream.cpp -ldl -lpthread -lz -o fastp_amd/libfastp_gpu_abl.so
echo built fastp_amd/libfastp_gpu_abl.so
