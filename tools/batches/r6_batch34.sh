#!/bin/bash
# same-box A/B: GPT-2-medium step with the bias column sums inside the grouped launch (new) vs separate column-sum passes (previous library); Bloom step (kernels unchanged)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$PWD/cleantransformer_amd/lib/variants/prev/libctmi355.so
for i in 1 2 3 4; do
  echo "== gpt2 new"; timeout 300 python tools/bench_gpt2.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  echo "== gpt2 prev"; CTMI_LIB_PATH=$V timeout 300 python tools/bench_gpt2.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
for i in 1 2 3; do
  echo "== bloom new"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  echo "== bloom prev"; CTMI_LIB_PATH=$V python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
