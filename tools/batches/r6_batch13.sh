#!/bin/bash
# kernel-boundary cost vs dirty L2 bytes and store policy (tools/probes/boundary_dirty.hip); GEMM parity of the unit-epilogue change; today's step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6b13
timeout 300 tools/probes/boundary_dirty.bin 2>&1 | tee gpurun_out/r6b13/boundary_dirty.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-padded-sample > gpurun_out/r6b13/bench.json 2> gpurun_out/r6b13/bench.err; tail -1 gpurun_out/r6b13/bench.json | cut -c1-300
