#!/bin/bash
# LM-head forward on the cross-lane (XLANE) 256-row kernel, with / without the early epilogue of the lagging row group, with / without nt stores
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=$PWD/cleantransformer_amd/lib/variants/lateepi/libctmi355.so
for i in 1 2; do
  echo "== shuffle epilogue (default)"; KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep "K=1024"
  echo "== xlane + early"; CTMI_XL_NT=1 KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep "K=1024"
  echo "== xlane + late"; CTMI_XL_NT=1 CTMI_LIB_PATH=$V KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep "K=1024"
  echo "== xlane + early, plain stores"; CTMI_GEMM_NT=0 KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep "K=1024"
  echo "== xlane + late, plain stores"; CTMI_GEMM_NT=0 CTMI_LIB_PATH=$V KS_LM=1 python tools/ksweep_probe.py 2>&1 | grep "K=1024"
done
