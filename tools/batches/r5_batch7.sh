#!/bin/bash
# fp16 compute dtype: op-level parity (GEMM, LayerNorm, attention, CE in half) and the model-level loops with dynamic loss scaling
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b7; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "float16 or dtype2" 2>&1 | tail -25 | tee $O/tests_ops_fp16.txt
timeout 900 python -m pytest tests/test_gpu_amp.py -q -m gpu 2>&1 | tail -30 | tee $O/tests_amp.txt
