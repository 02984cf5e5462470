#!/bin/bash
# fp16 WITH loss scaling (the reference's loop) against bf16, same box, interleaved
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b12; rm -rf $O; mkdir -p $O
for i in 1 2 3; do for d in fp16 bf16; do
timeout 200 python bench.py --dtype $d --no-cpu-baseline --no-padded-sample --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d step ms', d['ms_per_step'], 'loss', d.get('final_loss'), (d['roofline'].get('breakdown_ms_per_step') or {}).get('ms'))" | tee -a $O/bench_dtypes.txt
done; done
