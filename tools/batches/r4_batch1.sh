#!/bin/bash
# round-4 GPU batch 1: cold-vs-warm chain probe, then the load-phase variants (DMA first, static / no priorities) against the default
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b1; rm -rf $O; mkdir -p $O
timeout 600 python tools/chain_probe.py 24 2>&1 | grep -v amdgpu.ids | tee $O/chain_probe.txt
timeout 120 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -3 | tee $O/tests_default.txt
for v in dmafirst prio1 prio2; do
  CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -2 | tee $O/tests_$v.txt
done
for i in 1 2; do
  echo "== default" | tee -a $O/mb.txt; timeout 300 python tools/microbench.py gemm 2>&1 | grep -E "fwd|dgrad|wgrad" | tee -a $O/mb.txt
  for v in dmafirst prio1 prio2; do echo "== $v" | tee -a $O/mb.txt; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so timeout 300 python tools/microbench.py gemm 2>&1 | grep -E "fwd|dgrad|wgrad" | tee -a $O/mb.txt; done
done
for i in 1 2; do
  echo "== bench default" | tee -a $O/bench.txt; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt
  for v in dmafirst prio1 prio2; do echo "== bench $v" | tee -a $O/bench.txt; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tee -a $O/bench.txt; done
done
