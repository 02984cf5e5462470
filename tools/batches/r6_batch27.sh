#!/bin/bash
# grouped weight gradients: fp32 tiles stored in the accumulators' own layout (64-byte row pieces per instruction) and written through (variant grpnative) vs default
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$PWD/cleantransformer_amd/lib/variants/grpnative/libctmi355.so
CTMI_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_wgrad_grouped.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
  echo "== default"; timeout 300 python tools/microbench.py wgroup 2>&1 | grep grouped
  echo "== grpnative"; CTMI_LIB_PATH=$V timeout 300 python tools/microbench.py wgroup 2>&1 | grep grouped
done
for i in 1 2 3 4; do
  echo "== bench default"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== bench grpnative"; CTMI_LIB_PATH=$V python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
done
