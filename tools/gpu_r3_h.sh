#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or linear" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in nosteady default nosteady default; do
  if [ $v = default ]; then unset CTMI_LIB_PATH; else export CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so; fi
  echo "== $v"; timeout 300 python tools/microbench.py gemm 2>&1 | grep -v "amdgpu.ids\|wgrad"
done | tee $O/micro.log
unset CTMI_LIB_PATH
for v in nosteady default; do
  if [ $v = default ]; then unset CTMI_LIB_PATH; else export CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so; fi
  echo "== bench $v"; timeout 300 python bench.py --no-cpu-baseline --no-padded-sample --steps 10 | cut -c1-330
done | tee $O/bench.log
