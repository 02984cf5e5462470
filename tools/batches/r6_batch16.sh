#!/bin/bash
# GELU pre-activation written through (sc1, default build) vs non-temporal (variant auxnt): h->4h forward alone, step x5 interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$PWD/cleantransformer_amd/lib/variants/auxnt/libctmi355.so
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
echo "== auxwt"; timeout 300 python tools/microbench_wide.py 2>&1 | grep -i gelu
echo "== auxnt"; CTMI_LIB_PATH=$V timeout 300 python tools/microbench_wide.py 2>&1 | grep -i gelu
for i in 1 2 3 4 5; do
  echo "== bench auxwt"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== bench auxnt"; CTMI_LIB_PATH=$V python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
done
