#!/bin/bash
# lane plan on the GPU: parity subset, bench A/B (lane / split-scan / fused), kernel trace, SQ counters per kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
TAG=${1:-lane}
OUT=gpurun_out/lane_ab_$TAG.txt
: > $OUT
run() {  # name, env...
  NAME=$1; shift
  env "$@" timeout 300 python bench.py --steps 32 --warmup 8 --batches 8 --no-cpu > gpurun_out/ab_${TAG}_$NAME.log 2>&1
  python - "$NAME" gpurun_out/ab_${TAG}_$NAME.log >> $OUT <<'PY'
import sys, json
name, path = sys.argv[1:3]
try:
    j = json.loads(open(path).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"{name:34s} {j['value']:9.1f} Mreads/s  step {j['ms_per_step']:.3f} ms  kernels {r['kernel_avg_ms']:.4f} ms per {r['pairs_per_launch']} pairs  frac {r['frac']}")
except Exception as e:
    print(f"{name:34s} FAILED {e!r}: " + open(path).read()[-300:].replace("\n", " | "))
PY
  tail -1 $OUT
}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_option_fuzz.py -m gpu -x -q -k "not full_size and not file and not inflate and not deflate and not format and not parse and not evaluator" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$TAG.log
run lane_default FASTP_GPU_VERBOSE=1
run lane_nostats FASTP_GPU_DEBUG_SKIP=16
run lane_2wg FASTP_GPU_DEBUG_SKIP=16 FASTP_GPU_LANE_BLOCKS_PER_CU=2
run lane_3wg FASTP_GPU_DEBUG_SKIP=16 FASTP_GPU_LANE_BLOCKS_PER_CU=3
run scan_default FASTP_GPU_LANE=0
run fused FASTP_GPU_SPLIT=0
grep -h "fastp_gpu:" gpurun_out/ab_${TAG}_lane_default.log | head -1 >> $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${TAG}_trace -o trace -- python bench.py --steps 24 --warmup 2 --batches 4 --no-cpu > gpurun_out/rocprof_${TAG}.log 2>&1; echo "trace rc=$?"
python - gpurun_out/prof/${TAG}_trace >> $OUT <<'PY'
import sys, glob, csv
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    print("== " + f)
    for i, row in enumerate(csv.reader(open(f))):
        if i == 0 or row[0].startswith("fq_") or "fq_lane" in row[0]: print(",".join(x[:60] for x in row[:8]))
PY
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
P3="SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/${TAG}_sq$i -o pmc -- python bench.py --steps 1 --warmup 1 --batches 1 --no-cpu > gpurun_out/pmc_${TAG}_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
for K in _Z14fq_lane fq_stats; do echo "== SQ counters, $K (one launch of 4194304 pairs)" >> $OUT; python tools/pmc_parse.py $TAG $K >> $OUT; done
cat $OUT
