cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python tools/dup_forms_check.py 5 4194304 > gpurun_out/dup_forms.log 2>&1; echo "rc=$?"; grep -v amdgpu gpurun_out/dup_forms.log | tail -6
