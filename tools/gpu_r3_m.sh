#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3m; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.log
V=cleantransformer_amd/lib/variants/noesp/libctmi355.so
for i in 1 2; do
  echo "== noesp" | tee -a $O/epi.log; CTMI_LIB_PATH=$V timeout 100 python tools/microbench.py epi 2>&1 | grep -v amdgpu.ids | tee -a $O/epi.log
  echo "== esp" | tee -a $O/epi.log; timeout 100 python tools/microbench.py epi 2>&1 | grep -v amdgpu.ids | tee -a $O/epi.log
done
for i in 1 2 3; do
  CTMI_LIB_PATH=$V timeout 300 python bench.py --steps 20 --warmup 5 --no-breakdown --no-padded-sample --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 | tee -a $O/bench_noesp.log
  timeout 300 python bench.py --steps 20 --warmup 5 --no-breakdown --no-padded-sample --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 | tee -a $O/bench_esp.log
done
