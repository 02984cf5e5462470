#!/bin/bash
# (a) AdamW launch forms bit-identical; (b) the data-parallel launch policies at world 1: persistent / flow (new) / shared / reserve 16
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "adam or sgd" 2>&1 | tail -3
for i in 1 2 3; do
  for pol in 0 2 1; do
    echo "== CTMI_GEMM_SHARED=$pol"; CTMI_GEMM_SHARED=$pol python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  done
done
echo "== reserve16"; CTMI_GEMM_RESERVE_CUS=16 python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
