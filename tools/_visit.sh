cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/fuzz_more.py 1000 1500 > gpurun_out/fuzz_more_r02.log 2>&1; echo "fuzz rc=$?"; tail -5 gpurun_out/fuzz_more_r02.log
