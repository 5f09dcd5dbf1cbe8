#!/bin/bash
# round 6, visit f: the lane kernel's chunk hand-out from a counter in the workgroup's LDS (FASTP_GPU_LANE_DYNAMIC=2, the new
# default) against the global counter (1) and the static stride (0), full kernel and loads-only skeleton; --cut_front on the lane
# plan and the new GPU cases; the driver's command with every extra (the second stream now chosen by the overlap probe)
#   gpurun --timeout 2400 -- 'bash tools/gpu_r6_f.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6f}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
ABL="FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_abl.so BENCH_ALLOW_ABLATION=1"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "cut_front or joint_table or every_quality or plans_agree or at_baseline_scale or read_lengths or trim_and_cut or golden or (equals_oracle and not scale)" > gpurun_out/${V}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${V}_pytest.log
trace() {   # NAME ENV... : the lane / Stats kernels' averages of the bench's configuration
  NAME=$1; shift
  rm -rf gpurun_out/prof/${V}_$NAME
  env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${V}_$NAME -o t -- python bench.py --steps 16 --warmup 4 --batches 4 --no-cpu --no-extras > gpurun_out/${V}_$NAME.log 2>&1
  python - "$V" "$NAME" "$@" >> $OUT <<'PY'
import csv, glob, sys
v, name = sys.argv[1], sys.argv[2]
f = glob.glob(f"gpurun_out/prof/{v}_{name}/**/*kernel_stats.csv", recursive=True)
line = f"{name:22s} [{' '.join(a for a in sys.argv[3:] if 'FASTP_GPU_LIB' not in a and 'BENCH_ALLOW' not in a)}{' (profiling build)' if any('FASTP_GPU_LIB' in a for a in sys.argv[3:]) else ''}]"
if f:
    for r in csv.DictReader(open(f[0])):
        for key in ("fq_lane_kernel", "fq_stats5_kernel", "fq_stats_kernel", "fq_reduce_kernel"):
            if key in r["Name"]:
                line += f"  {key} {float(r['AverageNs'])/1e6:.4f} ms x{r['Calls']}"
print(line)
PY
  find gpurun_out/prof/${V}_$NAME -name "*_kernel_trace.csv" -delete
  tail -1 $OUT
}
trace local              FASTP_GPU_VERBOSE=1
trace global_grab4       FASTP_GPU_LANE_DYNAMIC=1
trace static             FASTP_GPU_LANE_DYNAMIC=0
trace skel_local         $ABL FASTP_GPU_DEBUG_SKIP=15
trace skel_static        $ABL FASTP_GPU_DEBUG_SKIP=15 FASTP_GPU_LANE_DYNAMIC=0
trace local_again        FASTP_GPU_VERBOSE=1
grep -h "second stream" gpurun_out/${V}_local.log | head -2 >> $OUT
cat $OUT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${V}_bench_driver_cmd.log 2> gpurun_out/${V}_bench_driver_cmd.err; echo "bench rc=$?"; tail -1 gpurun_out/${V}_bench_driver_cmd.log > gpurun_out/${V}_bench_driver_cmd.json; python - "$V" <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/{sys.argv[1]}_bench_driver_cmd.json").read())
print(j["value"], j["ms_per_step"], j["roofline"])
for r in j.get("other_configs", []): print({k: v for k, v in r.items() if k in ("config", "ms_per_step", "plan", "frac", "error")})
for k in ("cpu_baseline", "e2e_gpu", "e2e_dropin", "e2e_dropin_large"): print(k, {a: b for a, b in j.get(k, {}).items() if a not in ("what", "sample")})
PY
