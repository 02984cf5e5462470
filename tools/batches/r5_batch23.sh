#!/bin/bash
# cross-lane epilogue with whole 128-byte row pieces per store (rows r and r+8 trade halves by DPP): parity, then A/B against -DCTMI_XL_FULL_ROWS=0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5b23; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py -x -q -m gpu -k "gemm or linear or block" 2>&1 | tail -3 | tee $O/parity.txt
V=$PWD/cleantransformer_amd/lib/variants/halfrows/libctmi355.so
for i in 1 2; do
  echo "== logits on the cross-lane kernel, full rows"; CTMI_XL_NT=1 KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep "K=1024"
  echo "== logits on the cross-lane kernel, half rows"; CTMI_XL_NT=1 CTMI_LIB_PATH=$V KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep "K=1024"
  echo "== logits on the LDS-shuffled kernel (default)"; KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep "K=1024"
done | tee $O/lm.txt
bash tools/gpu_ab.sh "" "gemm" "qkv|dense|h4h|4hh" 3 2>&1 | grep -v wgrad | tee $O/ab.txt
