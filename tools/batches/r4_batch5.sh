#!/bin/bash
# round-4 GPU batch 5: cold anatomy of the fixed kernels; tile-rule and weight-gradient sweeps in the step (environment overrides)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b5; rm -rf $O; mkdir -p $O
timeout 600 python tools/gemm_anatomy.py 2>&1 | grep -v amdgpu.ids | tee $O/anatomy.txt
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  for e in "" "CTMI_TILE3_MIN=400" "CTMI_TILE3_MIN=600" "CTMI_WGRAD_NOSPLIT=2" "CTMI_WGRAD_NOSPLIT=1" "CTMI_WGRAD_STREAM=0"; do
    echo "== bench [$e]" | tee -a $O/bench.txt; env $e $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  done
done
