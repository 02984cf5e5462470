#!/bin/bash
# wave-quantisation tile rule (QKV forward onto the 128-row tile): env A/B on the same library, then GEMM parity
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b30; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3 4; do
  echo "== bench quant=1" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench quant=0" | tee -a $O/bench.txt; CTMI_TILE_QUANT=0 $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
done
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -3 | tee $O/tests.txt
