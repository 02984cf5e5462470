#!/bin/bash
# attention forward: two query blocks per workgroup, (last - i) then i (CTMI_ATTN_W32_PAIR=1, new default) vs one (=0): parity, kernel alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_attn_fuzz.py -x -q -m gpu -k "attn or attention" 2>&1 | tail -2
for i in 1 2 3; do
  echo "== pair"; timeout 300 python tools/microbench.py attn 2>&1 | grep "fwd\|bwd"
  echo "== single"; CTMI_ATTN_W32_PAIR=0 timeout 300 python tools/microbench.py attn 2>&1 | grep "fwd\|bwd"
done
