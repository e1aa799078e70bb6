#!/bin/bash
# sensitivity of the golden train-step cases to last-bit forward perturbations (which kernels run the same math)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_x}
mkdir -p $O
for v in default NO_ENTRY_CONV NO_BLOCKOUT_CONV1 NT_WSK_MIN_KT; do
  unset TUBER_NO_ENTRY_CONV TUBER_NO_BLOCKOUT_CONV1 TUBER_NT_WSK_MIN_KT
  case $v in NO_*) export TUBER_$v=1;; NT_WSK_MIN_KT) export TUBER_NT_WSK_MIN_KT=0;; esac
  python -m pytest tests/test_model_gpu.py -q -m gpu -s -k "test_train_step_matches_reference_golden" > $O/t_$v.log 2>&1
  echo "== $v rc $?"; grep "matcher assignments\|total loss hip\|global grad norm\|passed\|failed" $O/t_$v.log
done
