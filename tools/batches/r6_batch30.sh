#!/bin/bash
# GPT-2 path again (the padded dlogits registered on the autograd thread): parity, step new vs previous library, fused vs two-pass loss
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1800 python -m pytest tests/test_gpu_gpt.py -x -q -m gpu 2>&1 | tail -2
V=$PWD/cleantransformer_amd/lib/variants/prev/libctmi355.so
for i in 1 2 3; do
  echo "== new"; timeout 300 python tools/bench_gpt2.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  echo "== new, two-pass loss"; CTMI_FUSED_CE=0 timeout 300 python tools/bench_gpt2.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  echo "== previous library, two-pass loss"; CTMI_FUSED_CE=0 CTMI_LIB_PATH=$V timeout 300 python tools/bench_gpt2.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
