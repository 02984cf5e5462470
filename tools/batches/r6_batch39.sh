#!/bin/bash
# LM-head weight gradient: the tile rows of the last, 31 %-full round on the 128x256 tile in a second launch (default) vs one launch (CTMI_WGRAD_TAIL=0): parity, kernel, step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
for i in 1 2 3; do
  echo "== tail"; MB_ONLY=lm_head timeout 300 python tools/microbench.py gemm 2>&1 | grep "wgrad"
  echo "== one launch"; CTMI_WGRAD_TAIL=0 MB_ONLY=lm_head timeout 300 python tools/microbench.py gemm 2>&1 | grep "wgrad"
done
for i in 1 2 3 4; do
  echo "== bench tail"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== bench one launch"; CTMI_WGRAD_TAIL=0 python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
done
