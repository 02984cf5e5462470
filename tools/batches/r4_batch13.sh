#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b13; rm -rf $O; mkdir -p $O
python -c "
import ctypes as C, torch
torch.zeros(1, device='cuda')
hip=C.CDLL('libamdhip64.so'); lo,hi=C.c_int(0),C.c_int(0); print('rc',hip.hipDeviceGetStreamPriorityRange(C.byref(lo),C.byref(hi)),'least',lo.value,'greatest',hi.value)" | tee $O/range.txt
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
for i in 1 2 3; do
  for e in "" "CTMI_SIDE_STREAM_PRIORITY=low" "CTMI_SIDE_STREAM_PRIORITY=high"; do
    echo "== bench [$e]" | tee -a $O/bench.txt; env $e $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|Error.*' | tee -a $O/bench.txt
  done
done
