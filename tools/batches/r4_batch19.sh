#!/bin/bash
# K2 as the default build: GEMM / block / model parity (incl. the odd-K-step shapes), then the tile-3 threshold re-tuned on the same box
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b19; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py tests/test_gpu_bloom.py -x -q -m gpu -k "gemm or block or linear or bloom" 2>&1 | tail -3 | tee $O/tests.txt
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2; do
  for t in 350 600 1100 100000; do
    echo "== bench TILE3_MIN=$t" | tee -a $O/bench.txt; CTMI_TILE3_MIN=$t $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  done
done
for t in 350 600 1100 100000; do echo "== microbench TILE3_MIN=$t" >> $O/micro.txt; CTMI_TILE3_MIN=$t timeout 300 python tools/microbench.py gemm 2>&1 | grep -v lm_head | tail -13 >> $O/micro.txt; done
