#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3i; mkdir -p $O
for t in default 0 1 3 4; do
  if [ $t = default ]; then unset CTMI_GEMM_TILE; else export CTMI_GEMM_TILE=$t; fi
  echo "== tile $t"; timeout 300 python tools/microbench.py gemm epi 2>&1 | grep -v "amdgpu.ids\|lm_head"
done | tee $O/sweep.log
