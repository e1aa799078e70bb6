#!/bin/bash
# kernel tests of the three fused layer1 kernels, A/B bench lines (conv1 backward fused or not) and the kernel trace of the default build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_t}
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "conv4_bwd or conv1_bwd or blockout" > $O/t_k.log 2>&1; echo "kernel tests rc $?"; tail -2 $O/t_k.log
for v in fused no_conv1 fused2 no_conv1_2; do
  case $v in no_conv1*) export TUBER_NO_CONV1_BWD_FUSED=1;; *) unset TUBER_NO_CONV1_BWD_FUSED;; esac
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json,sys; d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])"
done
unset TUBER_NO_CONV1_BWD_FUSED
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
python scripts/rocpd_summary.py /tmp/kt/r_results.db 13 > $O/kernel_trace_stats.txt 2>&1
grep -n "conv1_bwd\|conv4_bwd\|blockout_conv1" $O/kernel_trace_stats.txt | cut -c1-160
