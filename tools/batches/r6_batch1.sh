#!/bin/bash
# round 6 baseline on today's pool: the round-5 tree's step, its non-GEMM kernels alone, and the breakdown
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6b1
python bench.py --no-cpu-baseline > gpurun_out/r6b1/bench.json 2> gpurun_out/r6b1/bench.err
tail -1 gpurun_out/r6b1/bench.json | cut -c1-1500
timeout 300 python tools/microbench.py attn ln ce adamw 2>&1 | tee gpurun_out/r6b1/microbench.txt | tail -40
