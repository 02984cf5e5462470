#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.log
V=cleantransformer_amd/lib/variants/r2rules/libctmi355.so
for i in 1 2; do
  CTMI_LIB_PATH=$V timeout 300 python bench.py --steps 20 --warmup 5 --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | tee -a $O/bench_old.log
  timeout 300 python bench.py --steps 20 --warmup 5 --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | tee -a $O/bench_new.log
done
