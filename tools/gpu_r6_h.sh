#!/bin/bash
# round 6, visit h: form 5 of the Stats kernel in column blocks (reads beyond 176 bases: configs[4]), the bench's extras with
# the per-run clear timed apart
#   gpurun --timeout 2400 -- 'bash tools/gpu_r6_h.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6h}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "every_quality or joint_table or config4 or read_lengths or adapter_fasta_on or (golden and not stream)" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
tracec() {   # NAME CONFIG ENV... : kernel averages of one line of other_configs
  NAME=$1; CFG=$2; shift; shift
  rm -rf gpurun_out/prof/${V}_$NAME
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_$NAME -o t -- python tools/one_config.py "$CFG" > gpurun_out/${V}_$NAME.log 2>&1
  python - "$V" "$NAME" "$@" >> $OUT <<'PY'
import csv, glob, sys
v, name = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/prof/{v}_{name}/**/*kernel_stats.csv", recursive=True)
print(f"== {name} {' '.join(sys.argv[3:])}")
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "at::native" not in r["Name"] and "elementwise" not in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:9]:
        n = r["Name"].split("(")[0].replace("void fq::", "").replace("void ", "")[:60]
        print(f"   {n:60s} avg {float(r['AverageNs'])/1e6:8.4f} ms  x{r['Calls']:>4s}  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
  grep '^{' gpurun_out/${V}_$NAME.log | cut -c1-260 >> $OUT
  find gpurun_out/prof/${V}_$NAME -name "*_kernel_trace.csv" -delete
}
tracec cfg4_form5  "configs[4]"   FASTP_GPU_VERBOSE=1
tracec cfg4_form4  "configs[4]"   FASTP_GPU_STATS_V=4
cat $OUT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${V}_bench_driver_cmd.log 2> gpurun_out/${V}_bench_driver_cmd.err; echo "bench rc=$?"; tail -1 gpurun_out/${V}_bench_driver_cmd.log > gpurun_out/${V}_bench_driver_cmd.json; python - "$V" <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/{sys.argv[1]}_bench_driver_cmd.json").read())
print(j["value"], j["ms_per_step"], j["roofline"])
for r in j.get("other_configs", []): print({k: v for k, v in r.items() if k in ("config", "ms_per_step", "reset_ms_per_run", "plan", "frac", "error")})
for k in ("cpu_baseline", "e2e_gpu", "e2e_dropin", "e2e_dropin_large", "e2e_dropin_bgzf"): print(k, {a: b for a, b in j.get(k, {}).items() if a not in ("what", "sample", "plain_gzip_inputs")})
PY
