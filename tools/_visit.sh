cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python tools/e2e_fastq.py --pairs 20000000 --threads 16 --gz-out > gpurun_out/e2e_r02.log 2>&1; echo "e2e rc=$?"; cat gpurun_out/e2e_r02.log | grep -v amdgpu.ids
