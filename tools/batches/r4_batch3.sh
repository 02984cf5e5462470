#!/bin/bash
# round-4 GPU batch 3: the K-loop vmcnt(0) fix and the side-tile LDS path — parity first, then same-box A/B (microbench, epilogue microbench, chain probe, step)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_block.py -x -q -m gpu -k "gemm or linear or block" 2>&1 | tail -4 | tee $O/tests.txt
VARS="oldwait noside ring5 ring6"
for i in 1 2; do
  echo "== default" | tee -a $O/mb.txt; timeout 300 python tools/microbench.py gemm epi 2>&1 | grep -E "fwd|dgrad|wgrad|res" | tee -a $O/mb.txt
  for v in $VARS; do echo "== $v" | tee -a $O/mb.txt; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so timeout 300 python tools/microbench.py gemm epi 2>&1 | grep -E "fwd|dgrad|wgrad|res" | tee -a $O/mb.txt; done
done
echo "== default" | tee -a $O/chain.txt; timeout 600 python tools/chain_probe.py 24 2>&1 | grep -v amdgpu.ids | tee -a $O/chain.txt
for v in oldwait ring5; do echo "== $v" | tee -a $O/chain.txt; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so timeout 600 python tools/chain_probe.py 24 2>&1 | grep -v amdgpu.ids | tee -a $O/chain.txt; done
for i in 1 2 3; do
  echo "== bench default" | tee -a $O/bench.txt; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  for v in $VARS; do echo "== bench $v" | tee -a $O/bench.txt; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt; done
done
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_default_full.json
