#!/bin/bash
# fewer fp32 slabs for the layer3 dW GEMMs (multi_reduce reads 1.1 GB of slabs per step): rows per slab 1408 (4 slabs) vs 2816 (2)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_z5}
mkdir -p $O
for r in 1408 1877 2816 5632; do
  echo "rows $r"; TUBER_TN_BIG_ROWS=$r python scripts/gemm_bench.py tngroup 2>&1 | grep "layer3\|layer4" | tee -a $O/tngroup.txt
done
for v in 1408 2816 1408b 2816b 1877; do
  export TUBER_TN_BIG_ROWS=${v%b}
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json,sys; d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print('rows $v', d['ms_per_step'])"
done
