cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "inflate or bgzf or deflate or pipeline" > gpurun_out/pytest_r02l.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_r02l.log
timeout 600 python tools/aux_bench.py 400000 > gpurun_out/aux_r02l.log 2>&1; echo "aux rc=$?"; tail -12 gpurun_out/aux_r02l.log
FASTP_GPU_INFLATE=lane timeout 600 python tools/aux_bench.py 400000 2>&1 | grep inflate
timeout 600 python tools/aux_bench.py 2000000 2>&1 | grep -i "flate\|ratio"
