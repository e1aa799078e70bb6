#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_h}
mkdir -p $O
TUBER_FORCE_DDP=1 rocprofv3 --kernel-trace --stats -d /tmp/kd -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/ddp_prof.log 2>&1
python scripts/rocpd_timeline.py /tmp/kd/r_results.db 14 > $O/ddp_timeline.txt 2>&1
python scripts/rocpd_gaps.py /tmp/kd/r_results.db 8 3 > $O/ddp_gaps.txt 2>&1
python scripts/rocpd_summary.py /tmp/kd/r_results.db 13 > $O/ddp_kernel_trace_stats.txt 2>&1
head -3 $O/ddp_timeline.txt; cat $O/ddp_gaps.txt | cut -c1-600
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "cfg1" -s > $O/t_cfg1.log 2>&1; echo "cfg1 rc $?"; grep "cfg1" $O/t_cfg1.log | head -3
python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; head -c 1500 $O/bench_default.json | cut -c1-1500
