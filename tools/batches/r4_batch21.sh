#!/bin/bash
# per-kernel times of the step: default (two stages landed before a K-step pair) vs LAND=1 (first K2 build, 2.6 ms faster) — which kernels differ?
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4b21; rm -rf $O; mkdir -p $O
VD=$PWD/cleantransformer_amd/lib/variants
B="python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample --steps 10 --warmup 3"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- $B > $O/default.json 2> $O/default.err
CTMI_LIB_PATH=$VD/land1/libctmi355.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_land1 -- $B > $O/land1.json 2> $O/land1.err
for d in default land1; do f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; done
find $O -name "*_kernel_trace.csv" -size +20M -delete
