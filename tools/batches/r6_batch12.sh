#!/bin/bash
# placement staggers: slab / scratch slots 4 KiB apart (default build) vs dense (variant nostagger); dlogits 4 KiB into its allocation vs aligned
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_block.py tests/test_gpu_bloom.py tests/test_gpu_gpt.py -x -q -m gpu -k "not bench_two_rank" 2>&1 | tail -2
V=$PWD/cleantransformer_amd/lib/variants/nostagger/libctmi355.so
for i in 1 2 3; do
  echo "== default (slots + dlogits staggered)"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  echo "== dense slots"; CTMI_LIB_PATH=$V python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  echo "== aligned dlogits"; CTMI_CE_STAGGER=0 python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  echo "== neither"; CTMI_CE_STAGGER=0 CTMI_LIB_PATH=$V python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
done
for c in 4096 0; do echo "== CE microbench, stagger $c"; CTMI_CE_STAGGER=$c python tools/microbench.py ce 2>&1 | grep -i "fused\|fwd\|bwd"; done
