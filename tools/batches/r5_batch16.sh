#!/bin/bash
# split DMA issue on the 256-row ping-pong tile (A pieces in the load phase, one / both B pieces after the MFMAs): parity on the variants, microbench, step
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
O=gpurun_out/r5b16; rm -rf $O; mkdir -p $O
for v in split1 split2; do
  echo "== parity $v"; CTMI_LIB_PATH=$R/cleantransformer_amd/lib/variants/$v/libctmi355.so timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or linear" 2>&1 | tail -2
done | tee $O/parity.txt
bash tools/gpu_ab.sh "" "gemm" "fwd|dgrad|wgrad" 3 2>&1 | tee $O/ab.txt
