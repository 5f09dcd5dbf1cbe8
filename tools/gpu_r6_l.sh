#!/bin/bash
# round 6, visit l: the text kernel's units in the fused claim, Stats form 5 with the trim's item on the masked path, the correction
# rounds taken apart (profiling build: 1024 no list entries, 2048 no edits, 4096 the mismatch words only)
#   gpurun --timeout 1800 -- 'bash tools/gpu_r6_l.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6l}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
ABL="FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_abl.so BENCH_ALLOW_ABLATION=1"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "exotic or text_kernel or sparse or stats or plans_agree or capacity or baseline_scale or dedup or test_gpu_equals_oracle" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
summ() {  # V NAME ARGS...
  python - "$@" >> $OUT <<'PY'
import csv, glob, sys, json
v, name = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/prof/{v}_{name}/**/*kernel_stats.csv", recursive=True)
line = f"{name:22s} [{' '.join(a for a in sys.argv[3:] if 'FASTP_GPU_LIB' not in a and 'BENCH_ALLOW' not in a)}{' (profiling build)' if any('FASTP_GPU_LIB' in a for a in sys.argv[3:]) else ''}]"
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "at::native" not in r["Name"] and "elementwise" not in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:7]:
        n = r["Name"].split("(")[0].replace("void fq::", "").replace("void ", "")[:44]
        line += f"  {n} {float(r['AverageNs'])/1e6:.4f} ms x{r['Calls']}"
print(line)
PY
  tail -1 $OUT | cut -c1-400
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
trace  headline           FASTP_GPU_VERBOSE=1
trace  headline_again     FASTP_GPU_VERBOSE=1
tracec softmask           "soft-masked"  FASTP_GPU_VERBOSE=1
tracec softmask_noclaim   "soft-masked"  FASTP_GPU_EXACT_CLAIM=0
tracec c_all              " -c "  $ABL
tracec c_nolist           " -c "  $ABL FASTP_GPU_DEBUG_SKIP=1024
tracec c_noedit           " -c "  $ABL FASTP_GPU_DEBUG_SKIP=2048
tracec c_noedit_nolist    " -c "  $ABL FASTP_GPU_DEBUG_SKIP=3072
tracec c_words_only       " -c "  $ABL FASTP_GPU_DEBUG_SKIP=4096
tracec c_nocorr           " -c "  $ABL FASTP_GPU_DEBUG_SKIP=32
cat $OUT
