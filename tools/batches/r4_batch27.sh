#!/bin/bash
# SPLIT2: 256-row ping-pong tiles issue their B pieces inside the MFMA phase (steady loop).  parity on the variant, GEMM microbench, step A/B
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b27; rm -rf $O; mkdir -p $O
V=$PWD/cleantransformer_amd/lib/variants/split2/libctmi355.so
CTMI_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm and not every_tile" 2>&1 | tail -3 | tee $O/tests_split2.txt
echo "== microbench default" > $O/micro.txt; timeout 300 python tools/microbench.py gemm 2>&1 | tail -16 >> $O/micro.txt
echo "== microbench split2" >> $O/micro.txt; CTMI_LIB_PATH=$V timeout 300 python tools/microbench.py gemm 2>&1 | tail -16 >> $O/micro.txt
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  echo "== bench default" | tee -a $O/bench.txt; $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench split2" | tee -a $O/bench.txt; CTMI_LIB_PATH=$V $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|bench.py: the loss.*' | tee -a $O/bench.txt
done
grep -E "==|qkv      fwd|h4h      fwd|lm_head" $O/micro.txt
