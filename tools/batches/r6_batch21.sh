#!/bin/bash
# block backward: sum of the K-halves + partial-row reductions in ONE launch at the end of the block (default) vs two launches (variant notail): parity, step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$PWD/cleantransformer_amd/lib/variants/notail/libctmi355.so
timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_wgrad_grouped.py tests/test_gpu_bloom.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3 4 5; do
  echo "== bench tail"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== bench notail"; CTMI_LIB_PATH=$V python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
done
