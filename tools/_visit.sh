cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipeline" > gpurun_out/pytest_r02m.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_r02m.log
for N in 400000 2000000; do for V in wave lane; do echo "== $N pairs, FASTP_GPU_INFLATE=$V"; FASTP_GPU_INFLATE=$V timeout 600 python tools/aux_bench.py $N 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/aux_r02m.log 2>&1
cat gpurun_out/aux_r02m.log
