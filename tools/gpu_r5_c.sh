#!/bin/bash
# round 5, second GPU visit: the Stats kernel's forms measured one against the other (per-kernel averages from rocprofv3
# --kernel-trace --stats, the same 4 batches of 4,194,304 pairs each), the ablations that give the measured floor of both
# kernels, SQ counters of the new default, the --phred64 cases + the new capacity test, and the N = 2 rehearsal on one GPU.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5_c.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
OUT=gpurun_out/r5c_stats_forms.txt
: > $OUT
timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -k "phred64 or cells_at_their_capacity or plans_agree or at_baseline_scale or work_list" > gpurun_out/r5c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5c_pytest.log
trace() {   # NAME ENV... : kernel averages of one configuration
  NAME=$1; shift
  rm -rf gpurun_out/prof/r5c_$NAME
  env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r5c_$NAME -o t -- python bench.py --steps 16 --warmup 4 --batches 4 --no-cpu --no-extras > gpurun_out/r5c_$NAME.log 2>&1
  python - "$NAME" "$@" >> $OUT <<'PY'
import csv, glob, sys, json
name = sys.argv[1]
f = glob.glob(f"gpurun_out/prof/r5c_{name}/**/*kernel_stats.csv", recursive=True)
line = f"{name:26s} [{' '.join(sys.argv[2:])}]"
if f:
    rows = {r["Name"]: r for r in csv.DictReader(open(f[0]))}
    for key in ("fq_lane_kernel", "fq_stats_kernel", "fq_reduce_kernel"):
        for n, r in rows.items():
            if key in n:
                line += f"  {key} {float(r['AverageNs'])/1e6:.4f} ms x{r['Calls']}"
try:
    j = json.loads(open(f"gpurun_out/r5c_{name}.log").read().strip().splitlines()[-1])
    line += f"  | step {j['ms_per_step']} ms, {j['value']} Mreads/s (under the tracer)"
except Exception as e:
    line += f"  | no bench line ({e})"
print(line)
PY
  find gpurun_out/prof/r5c_$NAME -name "*_kernel_trace.csv" -delete
}
trace v3_u64cells        FASTP_GPU_STATS_V=3
trace v4_kc4_hs32        FASTP_GPU_VERBOSE=1
trace v4_kc1_hs32        FASTP_GPU_STATS_KC=1
trace v4_kc2_hs32        FASTP_GPU_STATS_KC=2
trace v4_kc4_hs19        FASTP_GPU_STATS_HS=19
trace v4_kc4_hs20        FASTP_GPU_STATS_HS=20
trace v4_kc1_hs19        FASTP_GPU_STATS_KC=1 FASTP_GPU_STATS_HS=19
trace v4_kc4_hs32_1wg    FASTP_GPU_STATS_BLOCKS_PER_CU=1
trace v3_again           FASTP_GPU_STATS_V=3
# ---- the measured floor: what is left of each kernel when its steps are taken out one by one (FASTP_GPU_DEBUG_SKIP; results
# are meaningless then).  Stats: 64 no per-cycle adds, 128 no 5-mer adds, 256 no histogram adds.  Lane: 1 window predicate,
# 2 duplicate hash, 4 overlap analysis, 8 quality metrics.
trace skip_cyc           FASTP_GPU_DEBUG_SKIP=64
trace skip_kmer          FASTP_GPU_DEBUG_SKIP=128
trace skip_hist          FASTP_GPU_DEBUG_SKIP=256
trace skip_all_adds      FASTP_GPU_DEBUG_SKIP=448
trace lane_skip_window   FASTP_GPU_DEBUG_SKIP=1
trace lane_skip_hash     FASTP_GPU_DEBUG_SKIP=2
trace lane_skip_overlap  FASTP_GPU_DEBUG_SKIP=4
trace lane_skip_metrics  FASTP_GPU_DEBUG_SKIP=8
trace lane_loads_only    FASTP_GPU_DEBUG_SKIP=15
grep -h "stats kernel" gpurun_out/r5c_v4_kc4_hs32.log | head -2 >> $OUT
cat $OUT
# ---- SQ counters of the default form
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/r5c_sq$i -o pmc -- python bench.py --steps 1 --warmup 1 --batches 1 --no-cpu --no-extras > gpurun_out/r5c_pmc_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
(echo "== SQ counters, fq_stats_kernel, form 4 (u32 cells, KC 4, Hs 32), one launch of 4194304 pairs"; python tools/pmc_parse.py r5c fq_stats_kernel; echo "== fq_lane_kernel"; python tools/pmc_parse.py r5c fq_lane_kernel) > gpurun_out/r5c_sq_counters.txt
cat gpurun_out/r5c_sq_counters.txt
find gpurun_out/prof -name "*counter_collection.csv" -size +2M -delete
# ---- N = 2 on one GPU (gloo carries the collectives; the C ABI's RCCL path needs two devices): exchange / merge fields populated
BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --batches 3 --pairs 4194304 --no-extras --no-cpu > gpurun_out/r5c_n2.log 2>&1; echo "n2 rc=$?"
tail -1 gpurun_out/r5c_n2.log | cut -c1-2500
du -sh gpurun_out
