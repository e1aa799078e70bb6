#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_s}
mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1
python scripts/rocpd_summary.py /tmp/kt/r_results.db 13 > $O/kernel_trace_stats.txt 2>&1
python scripts/rocpd_sequence.py /tmp/kt/r_results.db 0 40 > $O/sequence.txt 2>&1
grep -n "conv1_bwd\|conv4_bwd\|blockout_conv1" $O/sequence.txt | cut -c1-140
