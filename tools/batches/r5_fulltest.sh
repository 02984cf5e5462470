#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5full; rm -rf $O; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -40 | tee $O/tests_all.txt
