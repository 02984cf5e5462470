#!/bin/bash
# ring drained before the epilogue's stores, the next two K-steps without a vmcnt wait (-DCTMI_PP_STORE_SKIP=1): parity on the variant, LM-head timings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$PWD/cleantransformer_amd/lib/variants/skipw/libctmi355.so
CTMI_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or linear" 2>&1 | tail -2
for i in 1 2 3; do
  echo "== default"; KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep "K=1024"; MB_ONLY=lm_head python tools/microbench.py gemm 2>&1 | grep -E "dgrad|wgrad"
  echo "== skipw"; CTMI_LIB_PATH=$V KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep "K=1024"; CTMI_LIB_PATH=$V MB_ONLY=lm_head python tools/microbench.py gemm 2>&1 | grep -E "dgrad|wgrad"
done
