#!/bin/bash
# long tiles (> 128 K-steps) one workgroup per tile under the persistent policy: step A/B (CTMI_GEMM_PERSIST_KSTEPS=0 = always persistent, the previous rule), parity of the GEMM tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3 4; do
  echo "== new rule"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  echo "== always persistent"; CTMI_GEMM_PERSIST_KSTEPS=0 python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or linear" 2>&1 | tail -2
