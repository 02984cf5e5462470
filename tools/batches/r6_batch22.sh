#!/bin/bash
# logits stores of the LM-head forward: nt (default) vs sc1 / sc1 nt / sc0 sc1 (write-through): kernel alone, step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  echo "== nt (default)"; MB_ONLY=lm_head MB_FWD_ONLY=1 timeout 300 python tools/microbench.py gemm 2>&1 | grep "lm_head"
  for v in lgsc1 lgsc1nt lgsys; do echo "== $v"; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so MB_ONLY=lm_head MB_FWD_ONLY=1 timeout 300 python tools/microbench.py gemm 2>&1 | grep "lm_head"; done
done
for i in 1 2 3; do
  echo "== bench nt"; python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo
  for v in lgsc1 lgsc1nt lgsys; do echo "== bench $v"; CTMI_LIB_PATH=$PWD/cleantransformer_amd/lib/variants/$v/libctmi355.so python bench.py --no-cpu-baseline --no-breakdown --no-padded-sample 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' | tr '\n' ' '; echo; done
done
