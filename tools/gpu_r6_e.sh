#!/bin/bash
# round 6, visit e: chunk hand-out sizes beyond 4, the lane kernel's skeleton with them (profiling build), the ablation of the
# EXT >= 2 instantiations (-c and --merge lines with steps left out), the driver's command with every extra
#   gpurun --timeout 2400 -- 'bash tools/gpu_r6_e.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6e}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
ABL="FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_abl.so BENCH_ALLOW_ABLATION=1"
summ() {  # V NAME ARGS...
  python - "$@" >> $OUT <<'PY'
import csv, glob, sys, json
v, name = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/prof/{v}_{name}/**/*kernel_stats.csv", recursive=True)
line = f"{name:22s} [{' '.join(a for a in sys.argv[3:] if 'FASTP_GPU_LIB' not in a and 'BENCH_ALLOW' not in a)}{' (profiling build)' if any('FASTP_GPU_LIB' in a for a in sys.argv[3:]) else ''}]"
if f:
    rows = list(csv.DictReader(open(f[0])))
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
trace grab4              FASTP_GPU_VERBOSE=1
trace grab1              FASTP_GPU_LANE_GRAB=1
trace grab8              FASTP_GPU_LANE_GRAB=8
trace grab16             FASTP_GPU_LANE_GRAB=16
trace skel_grab1         $ABL FASTP_GPU_DEBUG_SKIP=15 FASTP_GPU_LANE_GRAB=1
trace skel_grab4         $ABL FASTP_GPU_DEBUG_SKIP=15
trace skel_grab16        $ABL FASTP_GPU_DEBUG_SKIP=15 FASTP_GPU_LANE_GRAB=16
trace skel_static        $ABL FASTP_GPU_DEBUG_SKIP=15 FASTP_GPU_LANE_DYNAMIC=0
# the EXT >= 2 instantiations: what each step of the -c / --merge lane kernel costs
tracec c_all             "-c --cut_right"     FASTP_GPU_VERBOSE=1
tracec c_nocorr          "-c --cut_right"     $ABL FASTP_GPU_DEBUG_SKIP=32
tracec c_nowindow        "-c --cut_right"     $ABL FASTP_GPU_DEBUG_SKIP=1
tracec c_nooverlap       "-c --cut_right"     $ABL FASTP_GPU_DEBUG_SKIP=4
tracec c_nometrics       "-c --cut_right"     $ABL FASTP_GPU_DEBUG_SKIP=8
tracec c_skel            "-c --cut_right"     $ABL FASTP_GPU_DEBUG_SKIP=13
tracec m_all             "--merge --cut_right" FASTP_GPU_VERBOSE=1
tracec m_nocorr          "--merge --cut_right" $ABL FASTP_GPU_DEBUG_SKIP=32
tracec m_no2nd           "--merge --cut_right" $ABL FASTP_GPU_DEBUG_SKIP=64
tracec m_nooverlap       "--merge --cut_right" $ABL FASTP_GPU_DEBUG_SKIP=4
tracec f_all             "-f 5 -F 5"          FASTP_GPU_VERBOSE=1
tracec f_skel            "-f 5 -F 5"          $ABL FASTP_GPU_DEBUG_SKIP=13
cat $OUT | cut -c1-420
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${V}_bench_driver_cmd.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/${V}_bench_driver_cmd.log > gpurun_out/${V}_bench_driver_cmd.json; python - <<'PY'
import json
j = json.loads(open("gpurun_out/r6e_bench_driver_cmd.json").read())
print(j["value"], j["ms_per_step"], j["roofline"])
for r in j.get("other_configs", []): print(r)
for k in ("cpu_baseline", "e2e_gpu", "e2e_dropin", "e2e_dropin_large"): print(k, {a: b for a, b in j.get(k, {}).items() if a != "what"})
PY
