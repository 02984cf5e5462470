#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3o; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py tests/test_gpu_bloom.py tests/test_gpu_gpt.py tests/test_gpu_dropout.py tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.log
for i in 1 2; do
  W32_CASES=0,1 timeout 300 python tools/attn_w32_check.py check time 2>&1 | grep -v "amdgpu.ids\|worst\|max err" > $O/attn_$i.log; timeout 300 python bench.py --steps 20 --warmup 5 --no-padded-sample --no-cpu-baseline 2>/dev/null | tail -1 > $O/last.json
  python - <<'PY' | tee -a $O/ab.log
import json
d = json.load(open("gpurun_out/r3o/last.json"))
b = d["roofline"]["breakdown_ms_per_step"]["ms"]
print(d["ms_per_step"], d["value"], {k: round(v, 2) for k, v in b.items()})
PY
done
