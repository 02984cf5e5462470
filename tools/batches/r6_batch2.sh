#!/bin/bash
# round 6: new parity tests on the GPU (fp16 full-vocabulary dlogits, attention fuzz folded into -m gpu, drop-in import paths, dynamic-LDS refusal)
# + one kernel trace of the step (per-launch durations in order, for the AdamW / LayerNorm / attention launches)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6b2
timeout 1200 python -m pytest tests/test_gpu_attn_fuzz.py tests/test_gpu_dropin.py tests/test_gpu_amp.py -x -q -m gpu 2>&1 | tail -15
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6b2/trace -- python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample --steps 4 --warmup 2 > gpurun_out/r6b2/bench_traced.json 2> gpurun_out/r6b2/trace.err
find gpurun_out/r6b2/trace -name "*kernel_trace.csv" | head -2
