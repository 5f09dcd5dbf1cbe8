#!/bin/bash
# round 6, visit b: form 5 of the Stats kernel with the lane = (unit, item column) mapping and two lists per wavefront; its
# ablations (profiling build), occupancy variants, SQ counters of both kernels
#   gpurun --timeout 1500 -- 'bash tools/gpu_r6_b.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6b}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
ABL=FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_abl.so
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "joint_table or every_quality or cells_at_their_capacity or plans_agree or at_baseline_scale or work_list or read_lengths or (equals_oracle and not scale)" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
trace() {   # NAME ENV... : kernel averages of one configuration
  NAME=$1; shift
  rm -rf gpurun_out/prof/${V}_$NAME
  env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_$NAME -o t -- python bench.py --steps 16 --warmup 4 --batches 4 --no-cpu --no-extras > gpurun_out/${V}_$NAME.log 2>&1
  python - "$V" "$NAME" "$@" >> $OUT <<'PY'
import csv, glob, sys, json
v, name = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/prof/{v}_{name}/**/*kernel_stats.csv", recursive=True)
line = f"{name:22s} [{' '.join(a for a in sys.argv[3:] if 'FASTP_GPU_LIB' not in a)}{' (profiling build)' if any('FASTP_GPU_LIB' in a for a in sys.argv[3:]) else ''}]"
if f:
    rows = {r["Name"]: r for r in csv.DictReader(open(f[0]))}
    for key in ("fq_lane_kernel", "fq_stats_kernel", "fq_stats5_kernel", "fq_reduce_kernel"):
        for n, r in rows.items():
            if key in n:
                line += f"  {key} {float(r['AverageNs'])/1e6:.4f} ms x{r['Calls']}"
print(line)
PY
  find gpurun_out/prof/${V}_$NAME -name "*_kernel_trace.csv" -delete
  tail -1 $OUT
}
trace form5              FASTP_GPU_VERBOSE=1
trace form4              FASTP_GPU_STATS_V=4
trace form5_512          FASTP_GPU_STATS_THREADS=512
trace form5_768          FASTP_GPU_STATS_THREADS=768
trace form5_nocyc        $ABL FASTP_GPU_DEBUG_SKIP=64
trace form5_nokmer       $ABL FASTP_GPU_DEBUG_SKIP=128
trace form5_noadds       $ABL FASTP_GPU_DEBUG_SKIP=192
trace form5_again        FASTP_GPU_VERBOSE=1
grep -h "stats kernel" gpurun_out/${V}_form5.log | head -2 >> $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
P3="SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/${V}_sq$i -o pmc -- python bench.py --steps 1 --warmup 1 --batches 1 --no-cpu --no-extras > gpurun_out/${V}_pmc_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
for K in "fq_lane" fq_stats5; do echo "== SQ counters, $K (one launch of 4194304 pairs)" >> $OUT; python tools/pmc_parse.py $V "$K" >> $OUT; done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > gpurun_out/${V}_bench_driver_cmd.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/${V}_bench_driver_cmd.log | cut -c1-300
cat $OUT
