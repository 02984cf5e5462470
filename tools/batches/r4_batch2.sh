#!/bin/bash
# round-4 GPU batch 2: LDS-DMA issue probe; new parity tests (verbose margins); 5-stage ring variant in the cold chain
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b2; rm -rf $O; mkdir -p $O
timeout 120 tools/probes/dma_issue.bin 2>&1 | tee $O/dma_issue.txt
CTMI_TEST_VERBOSE=1 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -s -k "128row or step_shapes or bloom_attention" 2>&1 | grep -E "check_norm|step gemm|passed|failed|Error|error|assert" | tee $O/tests_new.txt | tail -80
for i in 1 2; do
  echo "== default"; timeout 600 python tools/chain_probe.py 24 2>&1 | grep -v amdgpu.ids
  echo "== ring5"; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/ring5/libctmi355.so timeout 600 python tools/chain_probe.py 24 2>&1 | grep -v amdgpu.ids
done | tee $O/chain_ring5.txt
for i in 1 2; do
  echo "== bench default" | tee -a $O/bench.txt; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  echo "== bench ring5" | tee -a $O/bench.txt; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/ring5/libctmi355.so python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
done
