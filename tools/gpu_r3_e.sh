#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3e; mkdir -p $O
timeout 600 python tools/attn_w32_check.py > $O/check.log 2>&1; echo "check rc=$?" | tee -a $O/check.log
grep -v "^OK" $O/check.log | tail -14
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py tests/test_gpu_bloom.py tests/test_gpu_gpt.py tests/test_gpu_dropout.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/bench.log
CTMI_ATTN_W32=0 timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_old.log 2>&1
tail -2 $O/bench.log | cut -c1-260; tail -1 $O/bench_old.log | cut -c1-260
