#!/bin/bash
# GPT-2 path: Conv1D forwards on the cross-lane 256-row kernel + the fused loss at the padded row pitch — parity (GPT-2, ops, Bloom), GPT-2-medium step vs the previous library (variant prev; the Python side of the fused loss is on in both: CTMI_FUSED_CE=0 turns it off)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1800 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_ops.py tests/test_gpu_bloom.py -x -q -m gpu 2>&1 | tail -2
V=$PWD/cleantransformer_amd/lib/variants/prev/libctmi355.so
for i in 1 2 3; do
  echo "== new"; timeout 300 python tools/bench_gpt2.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  echo "== new, two-pass loss"; CTMI_FUSED_CE=0 timeout 300 python tools/bench_gpt2.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  echo "== previous library"; CTMI_LIB_PATH=$V timeout 300 python tools/bench_gpt2.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
echo "== bloom step (unchanged kernels expected)"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
