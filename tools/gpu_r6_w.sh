#!/bin/bash
# round 6, visit w: does the drop-in's wall clock depend on the NUMA node its threads run on?
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python tools/numa_probe.py 12000000 > gpurun_out/r6w_numa_probe.txt 2> gpurun_out/r6w_numa_probe.err; echo "rc=$?"
cat gpurun_out/r6w_numa_probe.txt; tail -3 gpurun_out/r6w_numa_probe.err
