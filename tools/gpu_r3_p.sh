#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_dropout.py tests/test_gpu_gpt.py -x -q -m gpu 2>&1 | tail -4
