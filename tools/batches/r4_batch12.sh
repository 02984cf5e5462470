#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b12; rm -rf $O; mkdir -p $O
V=$PWD/cleantransformer_amd/lib/variants/nodbg/libctmi355.so
CTMI_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -2 | tee $O/tests.txt
for i in 1 2; do
  echo "== default" | tee -a $O/mb.txt; timeout 300 python tools/microbench.py gemm 2>&1 | grep -E "fwd|dgrad|wgrad" | tee -a $O/mb.txt
  echo "== nodbg" | tee -a $O/mb.txt; CTMI_LIB_PATH=$V timeout 300 python tools/microbench.py gemm 2>&1 | grep -E "fwd|dgrad|wgrad" | tee -a $O/mb.txt
done
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3 4; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench nodbg" | tee -a $O/bench.txt; CTMI_LIB_PATH=$V $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
done
