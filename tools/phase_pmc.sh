#!/bin/bash
# profiling only: the fused kernel with all but one dense phase left out (FASTP_GPU_DEBUG_SKIP), kernel time and SQ
# counters per configuration.  Results of such runs are meaningless as data - they size each phase's VALU / LDS load.
#   tools/phase_pmc.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
TAG=${1:-pp}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
for CFG in "all:0" "stats_only:47" "overlap_only:58" "hash_only:61" "masks_only:62" "metrics_only:55" "thin_only:31" "none:63"; do
  NAME=${CFG%%:*}; SKIP=${CFG##*:}
  export FASTP_GPU_DEBUG_SKIP=$SKIP
  timeout 300 python bench.py --steps 24 --warmup 4 --batches 4 --no-cpu > gpurun_out/pp_${TAG}_${NAME}.log 2>&1
  MS=$(tail -1 gpurun_out/pp_${TAG}_${NAME}.log | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_avg_ms'])" 2>/dev/null)
  i=0
  for P in "$P1" "$P2"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/${TAG}_${NAME}_sq$i -o pmc -- python bench.py --steps 1 --warmup 1 --batches 1 --no-cpu > /dev/null 2>&1
  done
  echo "== $NAME (skip=$SKIP): fq_fused_kernel ${MS} ms per 2,097,152 pairs"
  python tools/pmc_parse.py ${TAG}_${NAME} | grep -E "INSTS_VALU|INSTS_LDS |INSTS_LDS_ATOMIC|INSTS_SALU|LDS_IDX_ACTIVE|LDS_BANK_CONFLICT|WAIT_INST_LDS|LDS_ADDR"
done
