#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b17; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py -x -q -m gpu -k "gemm or block or linear" 2>&1 | tail -2 | tee $O/tests.txt
V=$PWD/cleantransformer_amd/lib/variants/auxplain/libctmi355.so
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3 4; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench auxplain" | tee -a $O/bench.txt; CTMI_LIB_PATH=$V $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
done
