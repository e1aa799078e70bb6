#!/bin/bash
# kernel-trace summary of graph replays only (bench.py --no-roofline): per-kernel totals + the phase timeline of the last step
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/qt
mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench.log 2>&1
python scripts/rocpd_summary.py /tmp/kt/r_results.db 13 > $O/kernel_trace_stats.txt 2>&1
python scripts/rocpd_timeline.py /tmp/kt/r_results.db 12 ${WIN:-0 30} ${WIN:-} > $O/timeline_graph_replay.txt 2>&1
python scripts/rocpd_context.py /tmp/kt/r_results.db ${CTX:-copyBuffer} 200 > $O/context.txt 2>&1
python scripts/rocpd_gaps.py /tmp/kt/r_results.db 15 8 > $O/gaps.txt 2>&1
tail -1 $O/bench.log | cut -c100-300
