#!/bin/bash
# bench.py's N > 1 control flow executed at FULL model size on a one-GPU box: two ranks on cuda:0 over gloo, patched in from outside by
# tests/bench_plumbing.py (bench.py has no such switch) — not a measurement
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5b25; rm -rf $O; mkdir -p $O
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tests/bench_plumbing.py 24,250880 --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > $O/out.json 2> $O/err.txt
echo "rc=$?"; python - <<'PY'
import json
for ln in open('gpurun_out/r5b25/out.json'):
    if ln.startswith('{'):
        d=json.loads(ln); print(d['metric'][:60], '| n_gpus', d['n_gpus'], '| ms_per_step', d['ms_per_step'], '| final_loss', d['final_loss']); print(d['config']['comm'])
PY
grep -E "comm candidate|Traceback" $O/err.txt | head
