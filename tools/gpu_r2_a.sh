#!/bin/bash
# round-2 GPU call A: new block-level path — parity tests, bench A/Bs, kernel-trace profile
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r2a; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_block.py -x -q -m gpu > $O/t_block.log 2>&1; echo "block rc=$?" >> $O/rc.txt
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
CTMI_WGRAD_STREAM=0 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_1stream.json 2>> $O/bench_default.err
CTMI_FUSED_CE=0 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_nofusedce.json 2>> $O/bench_default.err
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_bloom.py::test_config1_full_size_bf16_and_fp32_vs_oracle > $O/t_all.log 2>&1; echo "all rc=$?" >> $O/rc.txt
CTMI_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1stream -- python bench.py --no-cpu-baseline > $O/bench_1stream_under_rocprof.json 2> $O/prof_1stream.err
python tools/prof_sum.py $O/prof_1stream 13 > $O/prof_1stream_summary.txt 2>&1
timeout 300 python tools/microbench.py gemm attn ln ce adamw > $O/microbench.txt 2>&1
tail -3 $O/t_block.log; tail -3 $O/t_all.log; cat $O/rc.txt; cat $O/bench_default.json | cut -c1-600; cat $O/bench_1stream.json | cut -c1-300; cat $O/bench_nofusedce.json | cut -c1-300
find $O -name "*.csv" -size +3M -delete
