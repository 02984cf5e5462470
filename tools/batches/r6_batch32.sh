#!/bin/bash
# 256-row ping-pong tile, steady K-loop: half the waves of a row group issue their LDS-DMA pieces BEFORE their fragment reads (variant mixed) vs all reads first (default)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$PWD/cleantransformer_amd/lib/variants/mixed/libctmi355.so
CTMI_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
for i in 1 2; do
  echo "== default"; timeout 300 python tools/microbench.py gemm 2>&1 | grep "qkv      fwd\|h4h      fwd\|4hh      dgrad\|lm_head"
  echo "== mixed"; CTMI_LIB_PATH=$V timeout 300 python tools/microbench.py gemm 2>&1 | grep "qkv      fwd\|h4h      fwd\|4hh      dgrad\|lm_head"
done
