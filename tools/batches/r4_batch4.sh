#!/bin/bash
# round-4 GPU batch 4: side-tile LDS path v2 (3-step lead) vs the register prefetch, both with the K-loop vmcnt(0) fix
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py -x -q -m gpu -k "gemm or linear or block or fused_loss or ce_" 2>&1 | tail -3 | tee $O/tests.txt
VARS="noside oldwait"
for i in 1 2; do
  echo "== default" | tee -a $O/mb.txt; timeout 300 python tools/microbench.py epi 2>&1 | grep -E "fwd|dgrad|wgrad|res" | tee -a $O/mb.txt
  for v in $VARS; do echo "== $v" | tee -a $O/mb.txt; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so timeout 300 python tools/microbench.py epi 2>&1 | grep -E "fwd|dgrad|wgrad|res" | tee -a $O/mb.txt; done
done
for i in 1 2 3; do
  echo "== bench default" | tee -a $O/bench.txt; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  for v in $VARS; do echo "== bench $v" | tee -a $O/bench.txt; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt; done
done
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_default_full.json
CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/noside/libctmi355.so python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_noside_full.json
