cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof; export TMPDIR=/tmp
P1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_LDS"
for CFG in "all:0" "no_cyc:64" "no_kmer:128" "no_qh:256" "no_atomics:448" "no_stats:16"; do
  NAME=${CFG%%:*}; export FASTP_GPU_DEBUG_SKIP=${CFG##*:}
  MS=$(timeout 300 python bench.py --steps 24 --warmup 4 --batches 4 --no-cpu 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['kernel_avg_ms'])")
  timeout 300 rocprofv3 --pmc $P1 --output-format csv -d gpurun_out/prof/st_${NAME}_sq1 -o pmc -- python bench.py --steps 1 --warmup 1 --batches 1 --no-cpu > /dev/null 2>&1
  echo "== $NAME: fused ${MS} ms"; python tools/pmc_parse.py st_${NAME} | grep -E "INSTS_LDS |LDS_ATOMIC|IDX_ACTIVE|BANK_CONFLICT|ADDR_CONFLICT|INSTS_VALU"
done
