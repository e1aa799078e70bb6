#!/bin/bash
# round 3: kernel-trace of graph replays + the launch-by-launch listing of the whole shortest step
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_a}
mkdir -p $O
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
python scripts/rocpd_summary.py /tmp/kt/r_results.db 13 > $O/kernel_trace_stats.txt 2>&1
python scripts/rocpd_timeline.py /tmp/kt/r_results.db 12 0 30 > $O/timeline.txt 2>&1
python scripts/rocpd_sequence.py /tmp/kt/r_results.db 0 40 > $O/sequence.txt 2>&1
head -c 400 $O/bench.json
