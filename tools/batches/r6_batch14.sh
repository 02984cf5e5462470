#!/bin/bash
# write-through (sc1) stores of the large streamed outputs (GEMM epilogues, LayerNorm, 128-row attention) vs plain stores (variant stplain): parity, kernels alone, chain, step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6b14
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py tests/test_gpu_wgrad_grouped.py -x -q -m gpu 2>&1 | tail -3
V=$PWD/cleantransformer_amd/lib/variants/stplain/libctmi355.so
for i in 1 2; do
  echo "== wt (default)"; timeout 300 python tools/microbench.py gemm wgroup attn ln 2>&1 | grep -v amdgpu.ids | grep -v lm_head
  echo "== plain"; CTMI_LIB_PATH=$V timeout 300 python tools/microbench.py gemm wgroup attn ln 2>&1 | grep -v amdgpu.ids | grep -v lm_head
done
for i in 1 2; do
  echo "== chain wt"; timeout 300 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids | tail -8
  echo "== chain plain"; CTMI_LIB_PATH=$V timeout 300 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids | tail -8
done
for i in 1 2 3 4; do
  echo "== bench wt"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== bench plain"; CTMI_LIB_PATH=$V python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
done
