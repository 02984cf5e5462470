#!/bin/bash
# round 6: batch chains (CTMI_BLOCK_CHAINS=2: a block's launch chain issued as two chains over half the batch each, on two streams) — parity, then the step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CTMI_BLOCK_CHAINS=2 timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_bloom.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2 3; do
  for n in 1 2; do
    echo "== chains $n"; CTMI_BLOCK_CHAINS=$n python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  done
done
