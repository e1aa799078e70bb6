#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_z6}
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "bn_ or block_out or fold" > $O/t_k.log 2>&1; echo "kernel tests rc $?"; tail -2 $O/t_k.log
python -m pytest tests/test_model_gpu.py tests/test_training_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x > $O/t_m.log 2>&1; echo "model tests rc $?"; tail -2 $O/t_m.log
for v in a b; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json,sys; d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
done
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
python scripts/rocpd_summary.py /tmp/kt/r_results.db 13 > $O/kernel_trace_stats.txt 2>&1
grep -n "finalize\|stat_rows" $O/kernel_trace_stats.txt | cut -c1-170
