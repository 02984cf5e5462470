#!/bin/bash
# same box, interleaved: the round-5 tree (git 9107822, checked out under ab/r05 for this call only) vs the round-6 tree — python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3 4 5; do
  echo "== r06"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== r05"; (cd ab/r05 && python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '); echo
done
