#!/bin/bash
# kernel-trace summary of the eval forward in the default (eval precision) mode -> gpurun_out/et/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-et}
mkdir -p $O
EVAL_MODES=0 rocprofv3 --kernel-trace --stats -d /tmp/et -o r -- python scripts/eval_latency.py > $O/eval.log 2>&1
python scripts/rocpd_summary.py /tmp/et/r_results.db 47 > $O/kernel_trace_stats.txt 2>&1
python scripts/rocpd_sequence.py /tmp/et/r_results.db 0 40 > $O/sequence_last.txt 2>&1
tail -2 $O/eval.log
