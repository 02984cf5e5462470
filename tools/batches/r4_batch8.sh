#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b8; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bloom.py tests/test_gpu_gpt.py tests/test_gpu_trainer.py tests/test_gpu_amp.py tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -8 | tee $O/tests.txt
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  for e in "CTMI_WGRAD_DEFER_JOIN=0" ""; do
    echo "== bench [$e]" | tee -a $O/bench.txt; env $e $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' ' | tee -a $O/bench.txt; echo | tee -a $O/bench.txt
  done
done
