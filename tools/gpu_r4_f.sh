#!/bin/bash
# round 4: BASELINE configs[2] at its full size through the drop-in binding, against fastp_ref -w 1 (outputs + the whole JSON)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
df -h /dev/shm | tail -1
timeout 1500 python tools/e2e_dropin_100M.py --pairs 100000000 > gpurun_out/r04_dropin_100M_pairs.txt 2>&1; echo "100M drop-in rc=$?"
cat gpurun_out/r04_dropin_100M_pairs.txt | cut -c1-600
