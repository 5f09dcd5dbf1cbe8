#!/bin/bash
# round 5: the stream loop with the output copies handed to the writer thread (two device output sets): the stream / binding
# tests on the hardware, then the driver's bench command (its e2e legs time the drop-in on 4 M and 12 M pairs)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_stream_abi.py tests/test_ref_binding.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r5f_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5f_pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5f_bench_driver_cmd.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r5f_bench_driver_cmd.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], 'Mreads/s', j['ms_per_step'], 'ms/step', j['roofline']['frac'])
for k in j:
    if k.startswith('e2e'): print(k, json.dumps(j[k])[:700])
"
