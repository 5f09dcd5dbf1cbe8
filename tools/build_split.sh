#!/bin/bash
# The product library built file by file (objects under /tmp/fqobj, only what changed is compiled again): the same flags and the
# same sources as __graft_entry__.build()'s single hipcc command - fastp_gpu.hip is 6 of its 7 minutes - for the edit / measure loop.
#   tools/build_split.sh [NAME [-DFLAG ...]]  ->  fastp_amd/libfastp_gpu[_NAME].so
set -e
cd "$(dirname "$0")/.."
C=fastp_amd/csrc
NAME=$1; [ $# -gt 0 ] && shift
O=/tmp/fqobj${NAME:+_$NAME}
mkdir -p $O
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
pids=()
for f in fastp_gpu.hip fq_host.cpp fq_glue.cpp fq_comm.cpp fq_stream.cpp; do
  o=$O/${f%.*}.o
  if [ ! -f $o ] || [ -n "$(find $C include -newer $o \( -name '*.h' -o -name $f \) | head -1)" ]; then
    $HIPCC $FLAGS -c $C/$f -o $o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $O/*.o -ldl -lpthread -lz -o fastp_amd/libfastp_gpu${NAME:+_$NAME}.so
echo built fastp_amd/libfastp_gpu${NAME:+_$NAME}.so
