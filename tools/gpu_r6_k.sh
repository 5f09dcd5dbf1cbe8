#!/bin/bash
# round 6, visit k: the text kernel as a wavefront per unit (fq_text.h): its parity cases on the hardware, the option fuzz with
# every unit through it, the soft-masked line's kernels
#   gpurun --timeout 1500 -- 'bash tools/gpu_r6_k.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6k}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "exotic or text_kernel or sparse" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
FASTP_GPU_EXACT=1 timeout 600 python tools/fuzz_more.py 2000 2150 > gpurun_out/${V}_fuzz_text.log 2>&1; tail -4 gpurun_out/${V}_fuzz_text.log
tracec() {   # NAME CONFIG ENV... : kernel averages of one line of other_configs
  NAME=$1; CFG=$2; shift; shift
  rm -rf gpurun_out/prof/${V}_$NAME
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_$NAME -o t -- python tools/one_config.py "$CFG" > gpurun_out/${V}_$NAME.log 2>&1
  python - "$V" "$NAME" >> $OUT <<'PY'
import csv, glob, sys
v, name = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/prof/{v}_{name}/**/*kernel_stats.csv", recursive=True)
print(f"== {name}")
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "at::native" not in r["Name"] and "elementwise" not in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:12]:
        n = r["Name"].split("(")[0].replace("void fq::", "").replace("void ", "")[:60]
        print(f"   {n:60s} avg {float(r['AverageNs'])/1e6:8.4f} ms  x{r['Calls']:>4s}  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
  grep '^{' gpurun_out/${V}_$NAME.log | cut -c1-260 >> $OUT
  find gpurun_out/prof/${V}_$NAME -name "*_kernel_trace.csv" -delete
}
tracec softmask    "soft-masked"  FASTP_GPU_VERBOSE=1
tracec softmask_late "soft-masked" FASTP_GPU_EXACT_EARLY=0
tracec clean       "--adapter_sequence/" FASTP_GPU_VERBOSE=1
cat $OUT
