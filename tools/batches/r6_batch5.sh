#!/bin/bash
# AdamW: chunk-balanced launch (CTMI_ADAMW_FLAT) and staggered state buffers, on the real parameter set; then parity and the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  for f in 0 1; do for s in 0 1; do CTMI_ADAMW_FLAT=$f python tools/adamw_model_probe.py $s 2>&1 | grep AdamW; done; done
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_trainer.py tests/test_gpu_amp.py -x -q -m gpu -k "adam or Adam or optimizer or traj or resume or sgd or scal" 2>&1 | tail -3
for i in 1 2 3; do
  echo "== old"; CTMI_ADAMW_FLAT=0 python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  echo "== new"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
done
