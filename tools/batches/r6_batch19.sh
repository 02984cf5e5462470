#!/bin/bash
# dGELU data gradient on the 256x256 ping-pong tile with the side input prefetched a pass ahead (variant pre8; round 3 it spilled) vs the default (128x256 tile)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$PWD/cleantransformer_amd/lib/variants/pre8/libctmi355.so
for i in 1 2 3; do
  echo "== default"; timeout 300 python tools/microbench.py epi 2>&1 | grep "4hh dgrad"
  echo "== pre8"; CTMI_LIB_PATH=$V timeout 300 python tools/microbench.py epi 2>&1 | grep "4hh dgrad"
done
CTMI_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
