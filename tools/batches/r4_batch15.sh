#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b15; rm -rf $O; mkdir -p $O
V=$PWD/cleantransformer_amd/lib/variants/auxnt/libctmi355.so
CTMI_LIB_PATH=$V timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm_forward" 2>&1 | tail -1 | tee $O/tests.txt
for i in 1 2; do
  echo "== default"; timeout 300 python tools/chain_probe.py 24 2>&1 | grep -E "forward chain"
  echo "== auxnt"; CTMI_LIB_PATH=$V timeout 300 python tools/chain_probe.py 24 2>&1 | grep -E "forward chain"
done | tee $O/chain.txt
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3 4; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench auxnt" | tee -a $O/bench.txt; CTMI_LIB_PATH=$V $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
done
