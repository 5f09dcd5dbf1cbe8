#!/bin/bash
# round 5: the per-cycle table as v_mfma_i32_16x16x64_i8 (tools/microbench/stats_mfma.hip), timed and with its SQ counters
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
OUT=gpurun_out/r5e_stats_mfma.txt
(cd tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 stats_mfma.hip -o stats_mfma) 2>&1 | tail -2
tools/microbench/stats_mfma > $OUT 2>&1; echo "run rc=$?"
rm -rf gpurun_out/prof/r5e_sq
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/prof/r5e_sq -o pmc -- tools/microbench/stats_mfma 1048576 > gpurun_out/r5e_pmc.log 2>&1; echo "pmc rc=$?"
python - >> $OUT <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof/r5e_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mfma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== SQ counters of k_mfma, 1,048,576 reads x 152 cycles of one mate (average over its launches)")
for k in sorted(agg): print(f"{k:28s} n={len(agg[k])} avg={sum(agg[k])/len(agg[k]):.4g}")
cells = 152 * 1048576 / 64
if "SQ_INSTS_VALU" in agg: print(f"VALU wave-instructions per (cycle, 64 reads) cell group: {sum(agg['SQ_INSTS_VALU'])/len(agg['SQ_INSTS_VALU'])/cells:.2f}")
PY
cat $OUT
tail -3 gpurun_out/r5e_pmc.log
