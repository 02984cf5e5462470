#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or linear" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in gemmtiming_nosplit gemmtiming; do echo "== $v"; GEMM_VARIANT=$v timeout 300 python tools/gemm_anatomy.py 2>&1 | grep -v amdgpu.ids | cut -c1-200; done | tee $O/anatomy.log
for v in nosplit default nosplit default; do
  if [ $v = default ]; then unset CTMI_LIB_PATH; else export CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so; fi
  echo "== $v"; timeout 300 python tools/microbench.py gemm 2>&1 | grep -v amdgpu.ids
done | tee $O/micro.log
