#!/bin/bash
# whole-step hipGraph (cleantransformer_amd/graph.py): parity tests, then bench eager vs graphed
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -15
