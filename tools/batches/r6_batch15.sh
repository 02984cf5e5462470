#!/bin/bash
# write-through stores of the bf16 outputs only (fp32 weight gradients write-back again) vs plain stores: grouped launch alone, step x5 interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$PWD/cleantransformer_amd/lib/variants/stplain/libctmi355.so
echo "== wt"; timeout 300 python tools/microbench.py wgroup 2>&1 | grep grouped
echo "== plain"; CTMI_LIB_PATH=$V timeout 300 python tools/microbench.py wgroup 2>&1 | grep grouped
for i in 1 2 3 4 5; do
  echo "== bench wt"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== bench plain"; CTMI_LIB_PATH=$V python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
done
