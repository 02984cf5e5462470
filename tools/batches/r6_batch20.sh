#!/bin/bash
# dGELU data gradient on the 256-row ping-pong tile with the one-slot side prefetch (default) vs rounds 3-5 (variant pre4: forced onto the 128-row tile): parity, kernel, step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$PWD/cleantransformer_amd/lib/variants/pre4/libctmi355.so
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py tests/test_gpu_gpt.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
  echo "== pre8 (default)"; timeout 300 python tools/microbench.py epi 2>&1 | grep "4hh dgrad"
  echo "== pre4"; CTMI_LIB_PATH=$V timeout 300 python tools/microbench.py epi 2>&1 | grep "4hh dgrad"
done
for i in 1 2 3 4 5; do
  echo "== bench pre8"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== bench pre4"; CTMI_LIB_PATH=$V python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
done
