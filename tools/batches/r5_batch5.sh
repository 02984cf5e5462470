#!/bin/bash
# (a) parity hardening at full size: per-parameter gradients + bit-exact greedy decode vs the oracle; (b) what the data-parallel launch policies
# cost at world 1 on the final build (verdict item 5a): default / shared / reserve 16 / the round-4 DDP configuration
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bloom.py -x -q -s -m gpu -k "config1_full_size" 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/tests_fullsize.txt
B="bench.py --no-cpu-baseline --no-breakdown --no-padded-sample"
run() { n=$1; shift; for i in 1 2 3; do
  (env "$@" timeout 200 python $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'config': '$n', 'env': '$*', 'ms_per_step': d['ms_per_step'], 'final_loss': d.get('final_loss')}))" 2>&1) | tee -a $O/ddp_policy.jsonl; done; }
run default X=1
run shared CTMI_GEMM_SHARED=1
run reserve16 CTMI_GEMM_RESERVE_CUS=16
run shared_per_product_sidestream_join_per_block CTMI_GEMM_SHARED=1 CTMI_WGRAD_GROUP=0 CTMI_WGRAD_STREAM=1 CTMI_WGRAD_DEFER_JOIN=0
run per_product_sidestream_join_per_block CTMI_WGRAD_GROUP=0 CTMI_WGRAD_STREAM=1 CTMI_WGRAD_DEFER_JOIN=0
