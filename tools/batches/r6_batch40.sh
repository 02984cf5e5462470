#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "step_shapes" 2>&1 | tail -3
