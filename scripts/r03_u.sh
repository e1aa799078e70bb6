#!/bin/bash
# dW GEMM routing: layer4's down-sample weight gradient on the transpose-read kernels; the long-M class-branch pair on 128 x 128 tiles
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_u}
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "tn" > $O/t_tn.log 2>&1; echo "tn tests rc $?"; tail -2 $O/t_tn.log
echo DEFAULT; python scripts/gemm_bench.py tngroup 2>&1 | grep -v amdgpu.ids | tee $O/tngroup_default.txt
for r in 1408 2112 2816 4224; do
  echo "LONG rows $r"; TUBER_TN_BIG_MAX_M=20000 TUBER_TN_BIG_ROWS_LONG=$r python scripts/gemm_bench.py tngroup 2>&1 | grep "class-branch" | tee -a $O/tngroup_long.txt
done
for v in default long2112 default2 long2816; do
  case $v in long*) export TUBER_TN_BIG_MAX_M=20000 TUBER_TN_BIG_ROWS_LONG=${v#long};; *) unset TUBER_TN_BIG_MAX_M TUBER_TN_BIG_ROWS_LONG;; esac
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json,sys; d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
done
