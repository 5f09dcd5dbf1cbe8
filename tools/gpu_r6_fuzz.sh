#!/bin/bash
# round 6, last visit: the differential option fuzz on the hardware, final tree - more seeds than the suite runs, the switches' other
# positions included; the binding-level fuzz if time is left
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r6_fuzz_gpu.txt
: > $OUT
run() { echo "== $*" >> $OUT; env "$@" timeout 900 python tools/fuzz_more.py $FIRST $LAST 2>&1 | grep -v "^Hostname\|^Librccl\|version" >> $OUT; tail -1 $OUT; }
FIRST=200000 LAST=202500 run FASTP_GPU_VERBOSE=0
FIRST=203000 LAST=203600 run FASTP_GPU_EXACT=1
FIRST=204000 LAST=204600 run FASTP_GPU_LANE_POOL_LOG2=1
FIRST=205000 LAST=205600 run FASTP_GPU_MISC_FOLD_FIRST=0 FASTP_GPU_DUP_CLEAR_TAIL=0
FIRST=206000 LAST=206600 run FASTP_GPU_DUP_LOSERS_FIRST=1
cat $OUT
