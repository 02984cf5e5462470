#!/bin/bash
# Round evidence collection (run on the GPU box through gpurun): writes everything under gpurun_out/evidence/.
# rocprofv3 wants a writable TMPDIR; PMC counters go in their own passes (no trace domains beside --kernel-trace).
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/evidence; rm -rf $OUT; mkdir -p $OUT
R=${ROUND:-r06}
timeout 600 python bench.py --cpu-baseline ${CPU_BASELINE:-full} > $OUT/${R}_bench_default.json 2> $OUT/bench_default.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample --steps 10 --warmup 3 > $OUT/${R}_bench_under_rocprof.json 2> $OUT/prof_bench.err
# (round 5: the default runs the whole step on ONE stream — grouped weight gradients; the second trace is the round-4 form: four products per block on a side stream)
CTMI_WGRAD_GROUP=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench_1stream -- python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample --steps 10 --warmup 3 > $OUT/${R}_bench_per_product_under_rocprof.json 2> $OUT/prof_bench_1stream.err
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_step_MFMA -- python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample --steps 10 --warmup 3 > $OUT/pmc_step_MFMA.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_step_$c -- python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample --steps 10 --warmup 3 > $OUT/pmc_step_$c.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  MB_ONLY=lm_head MB_FWD_ONLY=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python tools/microbench.py gemm > $OUT/pmc_$c.log 2>&1
done
timeout 300 python tools/microbench.py gemm wgroup attn ln ce adamw > $OUT/${R}_microbench.txt 2>&1
timeout 300 python tools/energy_probe.py gemm wgroup attn hbm --seconds 1.0 > $OUT/${R}_energy_probe.txt 2>&1
timeout 200 python tools/microbench.py epi > $OUT/${R}_microbench_epilogues.txt 2>&1
timeout 200 python tools/attn_w32_check.py time > $OUT/${R}_attention_paths.txt 2>&1
timeout 300 python tools/bench_gpt2.py > $OUT/${R}_gpt2_medium_bench.txt 2>&1
timeout 400 python tools/bench_bloom7b1.py > $OUT/${R}_bloom7b1_1gpu_bench.txt 2>&1
timeout 120 python tools/blaslt_probe.py > $OUT/${R}_vendor_gemm_reference.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
ls -R $OUT | head -40
timeout 300 python tools/chain_probe.py 24 > $OUT/${R}_chain_probe.txt 2>&1
bash tools/pmc_gemm.sh > $OUT/${R}_pmc_lmhead_mfma.txt 2>&1
