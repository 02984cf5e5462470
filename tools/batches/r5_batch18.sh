#!/bin/bash
# the lagging row group's epilogue before its last barrier of the tile: parity (default build), then same-box A/B against -DCTMI_PP_EARLY_EPI=0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5b18; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wgrad_grouped.py -x -q -m gpu -k "gemm or linear or wgrad or grouped" 2>&1 | tail -3 | tee $O/parity.txt
for i in 1 2; do
  echo "== default"; KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep -v amdgpu
  echo "== lateepi"; KS_LM=1 CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/lateepi/libctmi355.so python tools/ksweep_probe.py 2>&1 | grep -v amdgpu
done | tee $O/ksweep.txt
bash tools/gpu_ab.sh "" "gemm wgroup" "fwd|dgrad|wgrad|grouped|separate" 3 2>&1 | tee $O/ab.txt
