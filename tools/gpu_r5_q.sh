#!/bin/bash
# round 5, visit q: instruction-cache counters of the lane kernel - the plain instantiation (90 KB of code) and the -c one (190 KB):
# the loop body of either is larger than the 64 KB instruction cache two CUs share
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
P="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU"
timeout 70 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/r5q_sq1 -o pmc -- python bench.py --steps 1 --warmup 1 --batches 1 --no-cpu --no-extras > gpurun_out/r5q_pmc_1.log 2>&1
echo "pmc pass 1 rc=$?"
timeout 70 rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/r5q_sq2 -o pmc -- python -c "
import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
for r in bench.other_configs(torch.device('cuda', 0), only='2x150 -c'): print(json.dumps(r))
" > gpurun_out/r5q_pmc_2.log 2>&1
echo "pmc pass 2 rc=$?"
(echo "== instruction cache, fq_lane_kernel<10,2,3,true,0> (driver's options, one launch of 4194304 pairs)"; python tools/pmc_parse.py r5q "fq_lane_kernel<10, 2, 3, true, 0>"
 echo "== fq_lane_kernel<10,2,3,true,2> (-c, launches of 2097152 pairs)"; python tools/pmc_parse.py r5q "fq_lane_kernel<10, 2, 3, true, 2>"
 echo "== fq_stats_kernel (both)"; python tools/pmc_parse.py r5q fq_stats_kernel) > gpurun_out/r5q_icache.txt
cat gpurun_out/r5q_icache.txt
find gpurun_out/prof -name "*counter_collection.csv" -size +2M -delete
