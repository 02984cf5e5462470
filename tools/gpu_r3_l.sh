#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3l; mkdir -p $O
MB_ONLY=h4h,4hh timeout 100 python tools/microbench.py gemm 2>&1 | grep wgrad | tee $O/mb.log
for i in 1 2; do
  for v in t0unsplit r2rules; do
  CTMI_LIB_PATH=cleantransformer_amd/lib/variants/$v/libctmi355.so timeout 300 python bench.py --steps 20 --warmup 5 --no-breakdown --no-padded-sample --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 | tee -a $O/bench_$v.log
  done
  timeout 300 python bench.py --steps 20 --warmup 5 --no-breakdown --no-padded-sample --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 | tee -a $O/bench_new.log
done
