#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3f; mkdir -p $O
timeout 600 python bench.py --cpu-baseline short --steps 20 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/bench.log
tail -3 $O/bench.log | cut -c1-3000
timeout 900 python -m pytest tests/test_gpu_bloom.py tests/test_gpu_gpt.py tests/test_gpu_decode.py tests/test_gpu_amp.py tests/test_gpu_trainer.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
