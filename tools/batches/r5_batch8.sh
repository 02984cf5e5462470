#!/bin/bash
# fp16: block-level and GPT-path tests, throughput of the functional path (timing only), the whole GPU suite
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b8; rm -rf $O; mkdir -p $O
timeout 200 python bench.py --dtype fp16 --no-cpu-baseline --no-padded-sample --steps 5 --warmup 2 2>$O/bench_fp16.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp16 step ms', d['ms_per_step'], 'loss', d.get('final_loss'), (d['roofline'].get('breakdown_ms_per_step') or {}).get('ms'))" | tee $O/bench_fp16.txt
tail -3 $O/bench_fp16.err
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/tests_all.txt
