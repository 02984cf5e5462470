#!/bin/bash
# (1) does the cold-operand race detector catch the K2 build whose prologue waited for one stage only (variant landp1)?  (2) full suite, smoke, bench on the default build
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b23; rm -rf $O; mkdir -p $O
VD=$PWD/cleantransformer_amd/lib/variants
CTMI_LIB_PATH=$VD/landp1/libctmi355.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "cold_operands" 2>&1 | tail -25 | tee $O/cold_on_racy_build.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench.json; python -c "import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['final_loss'])"
CTMI_LIB_PATH=$VD/landp1/libctmi355.so python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>&1 | tail -2 | cut -c1-300 | tee $O/bench_on_racy_build.txt
