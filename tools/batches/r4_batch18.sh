#!/bin/bash
# K2 (two ring stages per phase on the 128-row ping-pong tile): parity, GEMM microbench, step A/B
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b18; rm -rf $O; mkdir -p $O
V=$PWD/cleantransformer_amd/lib/variants/k2/libctmi355.so
CTMI_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py -x -q -m gpu -k "gemm or block or linear" 2>&1 | tail -3 | tee $O/tests_k2.txt
echo "== microbench default" > $O/micro.txt; timeout 300 python tools/microbench.py gemm 2>&1 | tail -40 >> $O/micro.txt
echo "== microbench k2" >> $O/micro.txt; CTMI_LIB_PATH=$V timeout 300 python tools/microbench.py gemm 2>&1 | tail -40 >> $O/micro.txt
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench k2" | tee -a $O/bench.txt; CTMI_LIB_PATH=$V $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
done
