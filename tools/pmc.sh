#!/bin/bash
# SQ counter passes over the bench workload (no --kernel-trace/--stats mixing: counters only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-pmc}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
P3="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/${TAG}_sq$i -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu > gpurun_out/pmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_parse.py $TAG
