#!/bin/bash
# round-5 verdict item 7: kernel-level evidence for the widened configs — GPT-2 medium (configs[3]) and the Bloom-7B1 geometry on one GPU (configs[4])
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6b7; rm -rf $O; mkdir -p $O
timeout 300 python tools/bench_gpt2.py > $O/r06_gpt2_medium_bench.txt 2> $O/gpt2.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gpt2 -- python tools/bench_gpt2.py --steps 5 --warmup 2 > $O/gpt2_under_rocprof.txt 2> $O/gpt2_prof.err
timeout 500 python tools/bench_bloom7b1.py > $O/r06_bloom7b1_1gpu_bench.txt 2> $O/7b1.err
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_7b1 -- python tools/bench_bloom7b1.py --steps 3 --warmup 1 > $O/7b1_under_rocprof.txt 2> $O/7b1_prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_gpt2_$c -- python tools/bench_gpt2.py --steps 3 --warmup 1 > $O/pmc_gpt2_$c.log 2>&1
  timeout 700 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_7b1_$c -- python tools/bench_bloom7b1.py --steps 2 --warmup 1 > $O/pmc_7b1_$c.log 2>&1
done
find $O -name "*_kernel_trace.csv" -size +30M -delete
find $O -name "*.csv" | head -30
cat $O/r06_gpt2_medium_bench.txt $O/r06_bloom7b1_1gpu_bench.txt | cut -c1-400
