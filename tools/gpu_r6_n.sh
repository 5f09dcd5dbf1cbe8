#!/bin/bash
# round 6, visit n: three compile-time A/Bs on ONE box (tools/build_ab.sh):
#   product = FQ_CORR_FLUSH 1, FQ_ST5_BOUNDARY 0, FQ_ST5_ONEBLK 1;  ab = FLUSH 0, BOUNDARY 1, ONEBLK 1;  ab2 = FLUSH 1, BOUNDARY 0, ONEBLK 0
#   gpurun --timeout 1800 -- 'bash tools/gpu_r6_n.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6n}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
AB="FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_ab.so"
AB2="FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_ab2.so"
env $AB timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "stats or plans_agree or baseline_scale or test_gpu_equals_oracle or corr" > gpurun_out/${V}_pytest_ab.log 2>&1; echo "pytest (ab library) rc=$?"; tail -3 gpurun_out/${V}_pytest_ab.log
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "stats or plans_agree or baseline_scale or corr" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest (product) rc=$?"; tail -3 gpurun_out/${V}_pytest.log
summ() {  # V NAME ARGS...
  python - "$@" >> $OUT <<'PY'
import csv, glob, sys, json
v, name = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/prof/{v}_{name}/**/*kernel_stats.csv", recursive=True)
line = f"{name:22s} [{' '.join(a.split('/')[-1] for a in sys.argv[3:])}]"
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "at::native" not in r["Name"] and "elementwise" not in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:6]:
        n = r["Name"].split("(")[0].replace("void fq::", "").replace("void ", "")[:44]
        line += f"  {n} {float(r['AverageNs'])/1e6:.4f} ms x{r['Calls']}"
print(line)
PY
  tail -1 $OUT | cut -c1-330
}
trace() {   # NAME ENV... : kernel averages of the bench's configuration
  NAME=$1; shift
  rm -rf gpurun_out/prof/${V}_$NAME
  env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_$NAME -o t -- python bench.py --steps 16 --warmup 4 --batches 4 --no-cpu --no-extras > gpurun_out/${V}_$NAME.log 2>&1
  summ "$V" "$NAME" "$@"
  find gpurun_out/prof/${V}_$NAME -name "*_kernel_trace.csv" -delete
}
tracec() {   # NAME CONFIG ENV... : kernel averages of one line of other_configs
  NAME=$1; CFG=$2; shift; shift
  rm -rf gpurun_out/prof/${V}_$NAME
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_$NAME -o t -- python tools/one_config.py "$CFG" > gpurun_out/${V}_$NAME.log 2>&1
  summ "$V" "$NAME" "$@"
  grep '^{' gpurun_out/${V}_$NAME.log | cut -c1-200 >> $OUT
  find gpurun_out/prof/${V}_$NAME -name "*_kernel_trace.csv" -delete
}
trace  head_product       FASTP_GPU_VERBOSE=1
trace  head_ab_boundary   $AB
trace  head_ab2_blocks    $AB2
trace  head_product_2     FASTP_GPU_VERBOSE=1
trace  head_ab_boundary_2 $AB
trace  head_ab2_blocks_2  $AB2
tracec c_flush            " -c "   FASTP_GPU_VERBOSE=1
tracec c_per_round        " -c "   $AB
tracec c_flush_2          " -c "   FASTP_GPU_VERBOSE=1
tracec c_per_round_2      " -c "   $AB
tracec m_flush            "--merge"  FASTP_GPU_VERBOSE=1
tracec m_per_round        "--merge"  $AB
cat $OUT
