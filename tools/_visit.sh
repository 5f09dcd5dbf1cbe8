cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/aux_bench.py 400000 2>&1 | grep -i "deflate\|ratio"
timeout 600 python tools/aux_bench.py 2000000 2>&1 | grep -i "deflate\|ratio"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deflate or all_streams_and_gzip" 2>&1 | tail -3
