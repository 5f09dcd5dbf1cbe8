#!/bin/bash
# Stats kernel variants: padded class stride on/off; quick parity first
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-st3}
OUT=gpurun_out/stats_ab_$TAG.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "equals_oracle or read_lengths or at_scale" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$TAG.log
run() { NAME=$1; shift; env "$@" timeout 300 python bench.py --steps 32 --warmup 8 --batches 8 --no-cpu --no-extras > gpurun_out/ab_${TAG}_$NAME.log 2>&1; tail -1 gpurun_out/ab_${TAG}_$NAME.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$NAME', j['value'], 'Mreads/s kernels', r['kernel_avg_ms'], 'ms per', r['pairs_per_launch'], r.get('kernels'))" | tee -a $OUT; }
run pad_default FASTP_GPU_VERBOSE=1
run pad_off FASTP_GPU_STATS_PAD=0
run pad_default_b
run pad_off_b FASTP_GPU_STATS_PAD=0
run pad_1wg FASTP_GPU_STATS_BLOCKS_PER_CU=1
grep -h "stats" gpurun_out/ab_${TAG}_pad_default.log | head -3 >> $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/${TAG}_sq$i -o pmc -- python bench.py --steps 1 --warmup 1 --batches 1 --no-cpu --no-extras > gpurun_out/pmc_${TAG}_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
echo "== SQ counters, fq_stats (one launch of 4194304 pairs)" >> $OUT; python tools/pmc_parse.py $TAG fq_stats >> $OUT
cat $OUT
