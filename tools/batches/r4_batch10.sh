#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b10; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py -x -q -m gpu -k "gemm or block" 2>&1 | tail -3 | tee $O/tests.txt
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  for e in "CTMI_WGRAD_RULE=0" "" "CTMI_WGRAD_RULE=2" "CTMI_WGRAD_RULE=2 CTMI_WGRAD_ITEMS4=256" "CTMI_WGRAD_RULE=2 CTMI_WGRAD_ITEMS4=64"; do
    echo "== bench [$e]" | tee -a $O/bench.txt; env $e $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  done
done
