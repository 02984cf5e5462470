#!/bin/bash
# compile-time variants re-measured on the round-6 step: GELUG (forward saves gelu'(u), backward multiplies) and the lazy rescale of the attention forward
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in gelug defer8; do
  echo "== parity $v"; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so timeout 600 python -m pytest tests/test_gpu_block.py tests/test_gpu_bloom.py -x -q -m gpu -k "not bench_two_rank" 2>&1 | tail -2
done
for i in 1 2 3; do
  echo "== bench default"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  for v in gelug defer8; do echo "== bench $v"; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done
done
