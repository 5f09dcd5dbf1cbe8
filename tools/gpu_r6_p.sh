#!/bin/bash
# round 6, visit p: Stats form 5 with the reads' last column out of the lane mapping (product) against -DFQ_ST5_TAILCOL=0 (ab4), same
# box; the -c / --merge steps as ONE launch (the correction list just under 2^30 entries) against FASTP_GPU_CORR_LIST_LOG2=29
#   gpurun --timeout 1800 -- 'bash tools/gpu_r6_p.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6p}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
AB4="FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_ab4.so"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "stats or plans_agree or baseline_scale or test_gpu_equals_oracle or corr or merge" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest (product) rc=$?"; tail -3 gpurun_out/${V}_pytest.log
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
trace  head_tailcol       FASTP_GPU_VERBOSE=1
trace  head_allcols       $AB4
trace  head_tailcol_2     FASTP_GPU_VERBOSE=1
trace  head_allcols_2     $AB4
tracec c_one_launch       " -c "     FASTP_GPU_VERBOSE=1
tracec c_two_launches     " -c "     FASTP_GPU_CORR_LIST_LOG2=29
tracec m_one_launch       "--merge"  FASTP_GPU_VERBOSE=1
tracec m_two_launches     "--merge"  FASTP_GPU_CORR_LIST_LOG2=29
tracec c_one_launch_2     " -c "     FASTP_GPU_VERBOSE=1
tracec c_two_launches_2   " -c "     FASTP_GPU_CORR_LIST_LOG2=29
cat $OUT
