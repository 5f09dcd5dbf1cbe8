#!/bin/bash
# lane kernel stage ablations + counters
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
TAG=${1:-lane2}
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
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "equals_oracle or stress or read_lengths or launches or golden" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_$TAG.log
run lane_default FASTP_GPU_VERBOSE=1
run lane_nostats FASTP_GPU_DEBUG_SKIP=16
run lane_only_load FASTP_GPU_DEBUG_SKIP=31
run lane_no_window FASTP_GPU_DEBUG_SKIP=17
run lane_no_hash FASTP_GPU_DEBUG_SKIP=18
run lane_no_overlap FASTP_GPU_DEBUG_SKIP=20
run lane_no_metrics FASTP_GPU_DEBUG_SKIP=24
run lane_2wg FASTP_GPU_DEBUG_SKIP=16 FASTP_GPU_LANE_BLOCKS_PER_CU=2
run lane_1wg FASTP_GPU_DEBUG_SKIP=16 FASTP_GPU_LANE_BLOCKS_PER_CU=1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
P3="SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/${TAG}_sq$i -o pmc -- python bench.py --steps 1 --warmup 1 --batches 1 --no-cpu > gpurun_out/pmc_${TAG}_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
for K in "void fq_lane" fq_stats; do echo "== SQ counters, $K (one launch of 4194304 pairs)" >> $OUT; python tools/pmc_parse.py $TAG "$K" >> $OUT; done
cat $OUT
# variant: the lane kernel compiled for 4 wavefronts per SIMD (128 VGPRs)
if [ -f fastp_amd/libfastp_gpu_w4.so ]; then
  run() { NAME=$1; shift; env "$@" timeout 300 python bench.py --steps 32 --warmup 8 --batches 8 --no-cpu > gpurun_out/ab_${TAG}_$NAME.log 2>&1; tail -1 gpurun_out/ab_${TAG}_$NAME.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$NAME', j['value'], 'Mreads/s kernels', r['kernel_avg_ms'], 'ms per', r['pairs_per_launch'])" | tee -a $OUT; }
  run w4_default FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_w4.so
  run w4_nostats FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_w4.so FASTP_GPU_DEBUG_SKIP=16
fi
