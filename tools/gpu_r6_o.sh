#!/bin/bash
# round 6, visit o: the no-front instantiation of Stats form 5 (product) against the one that reads the front (ab3 = -DFQ_ST5_NOFRONT=0),
# same box; the Stats kernel's parity cases on the final form; the driver's command
#   gpurun --timeout 1800 -- 'bash tools/gpu_r6_o.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6o}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
AB3="FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_ab3.so"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "stats or plans_agree or baseline_scale or test_gpu_equals_oracle or front or umi" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest (product) rc=$?"; tail -3 gpurun_out/${V}_pytest.log
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
trace  head_nofront       FASTP_GPU_VERBOSE=1
trace  head_front_read    $AB3
trace  head_nofront_2     FASTP_GPU_VERBOSE=1
trace  head_front_read_2  $AB3
cat $OUT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${V}_bench_driver_cmd.log 2> gpurun_out/${V}_bench_driver_cmd.err; echo "bench rc=$?"; tail -1 gpurun_out/${V}_bench_driver_cmd.log > gpurun_out/${V}_bench_driver_cmd.json; python - "$V" <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/{sys.argv[1]}_bench_driver_cmd.json").read())
print(j["value"], j["ms_per_step"], j["roofline"])
for r in j.get("other_configs", []): print({k: v for k, v in r.items() if k in ("config", "ms_per_step", "plan", "frac", "error")})
for k in ("cpu_baseline", "e2e_gpu", "e2e_dropin", "e2e_dropin_large", "e2e_dropin_bgzf"): print(k, {a: b for a, b in j.get(k, {}).items() if a not in ("what", "sample", "plain_gzip_inputs")})
PY
