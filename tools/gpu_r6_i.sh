#!/bin/bash
# round 6, visit i: SQ counters of configs[4]'s kernels (fq_ovr_count_kernel is the step's critical path behind the lane kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=r6i
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_WAVES SQ_ACTIVE_INST_SCA"
P3="SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INSTS_GDS"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/${V}_sq$i -o pmc -- python tools/one_config.py "configs[4]" > gpurun_out/${V}_pmc_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
for K in fq_ovr_count fq_ovr_tasks "fq_lane" fq_stats5; do echo "== SQ counters, $K (configs[4]: 2 M pairs of 2x250 per launch)"; python tools/pmc_parse.py $V "$K"; done > gpurun_out/${V}_sq.txt
cat gpurun_out/${V}_sq.txt
