#!/bin/bash
# fp16 on the fast kernel families (LDS-DMA GEMMs, grouped weight gradients, 128-row attention): tests, then its step time; bf16 default re-checked
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b9; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wgrad_grouped.py tests/test_gpu_amp.py tests/test_gpu_block.py tests/test_gpu_gpt.py -q -m gpu -k "float16 or fp16 or half or dtype2 or 128row or one_call" 2>&1 | tail -15 | tee $O/tests_fp16.txt
for d in fp16 bf16; do
timeout 200 python bench.py --dtype $d --no-cpu-baseline --no-padded-sample --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d step ms', d['ms_per_step'], 'loss', d.get('final_loss'), (d['roofline'].get('breakdown_ms_per_step') or {}).get('ms'))" | tee -a $O/bench_dtypes.txt
done
