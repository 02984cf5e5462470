#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3n; rm -rf $O; mkdir -p $O
for i in 1 2; do
  for v in t0unsplit noesp cur; do
    if [ $v = cur ]; then unset CTMI_LIB_PATH; else export CTMI_LIB_PATH=cleantransformer_amd/lib/variants/$v/libctmi355.so; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-padded-sample --no-cpu-baseline 2>/dev/null | tail -1 > $O/last.json
    python - $v <<'PY' | tee -a $O/ab.log
import json, sys
d = json.load(open("gpurun_out/r3n/last.json"))
b = d["roofline"]["breakdown_ms_per_step"]["ms"]
print(sys.argv[1], d["ms_per_step"], "lmfwd", d["roofline"]["largest_launch"]["avg_launch_ms"], {k: round(v, 2) for k, v in b.items()})
PY
  done
done
