#!/bin/bash
# round 6, visit d: the lane kernel's chunk hand-out (FASTP_GPU_LANE_GRAB chunks per returning atomic) and its row prefetch into
# L2 at 128 / 64 / 32-byte steps (FASTP_GPU_LANE_PREFETCH), each against the default and on the loads-only skeleton of the
# profiling build; then the stream-overlap probe (tools/stream_overlap_probe.py)
#   gpurun --timeout 1500 -- 'bash tools/gpu_r6_d.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
V=${1:-r6d}
OUT=gpurun_out/${V}_ab.txt
: > $OUT
ABL=FASTP_GPU_LIB=$PWD/fastp_amd/libfastp_gpu_abl.so
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
trace default            FASTP_GPU_VERBOSE=1
trace grab2              FASTP_GPU_LANE_GRAB=2
trace grab4              FASTP_GPU_LANE_GRAB=4
trace pf3_128            FASTP_GPU_LANE_PREFETCH=3
trace pf3_64             FASTP_GPU_LANE_PREFETCH=7
trace pf3_32             FASTP_GPU_LANE_PREFETCH=11
trace pf1_64             FASTP_GPU_LANE_PREFETCH=5
trace grab2_pf3_64       FASTP_GPU_LANE_GRAB=2 FASTP_GPU_LANE_PREFETCH=7
trace skel               $ABL FASTP_GPU_DEBUG_SKIP=15
trace skel_grab2         $ABL FASTP_GPU_DEBUG_SKIP=15 FASTP_GPU_LANE_GRAB=2
trace skel_grab4         $ABL FASTP_GPU_DEBUG_SKIP=15 FASTP_GPU_LANE_GRAB=4
trace skel_grab4_pf64    $ABL FASTP_GPU_DEBUG_SKIP=15 FASTP_GPU_LANE_GRAB=4 FASTP_GPU_LANE_PREFETCH=7
trace skel_static        $ABL FASTP_GPU_DEBUG_SKIP=15 FASTP_GPU_LANE_DYNAMIC=0
trace default_again      FASTP_GPU_VERBOSE=1
cat $OUT
timeout 600 python tools/stream_overlap_probe.py > gpurun_out/${V}_overlap_probe.log 2>&1; echo "probe rc=$?"; grep '^{' gpurun_out/${V}_overlap_probe.log
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/stream_overlap_probe.py "soft-masked" > gpurun_out/${V}_overlap_probe_q8.log 2>&1; echo "probe q8 rc=$?"; grep '^{' gpurun_out/${V}_overlap_probe_q8.log
