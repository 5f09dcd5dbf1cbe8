cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ref_binding.py tests/test_abi.py -m gpu -x -q -k "overrep or abi or pe_default or se_adapter" 2>&1 | tail -4
