#!/bin/bash
# grouped weight gradients with the column sums of the B operand ([in,out] weights: GPT-2's bias gradients without separate column-sum passes): parity, GPT-2-medium step vs CTMI_WGRAD_GROUP=0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1800 python -m pytest tests/test_gpu_wgrad_grouped.py tests/test_gpu_block.py tests/test_gpu_gpt.py -x -q -m gpu 2>&1 | tail -15
