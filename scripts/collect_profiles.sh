#!/bin/bash
# Runs on the GPU box (via gpurun): default bench line, rocprofv3 kernel-trace stats of the same command, and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs, no other trace domains) -> small text summaries under gpurun_out/fin/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fin
mkdir -p $O
python bench.py > $O/bench_default.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
python scripts/rocpd_summary.py /tmp/kt/r_results.db 13 > $O/kernel_trace_stats.txt 2>&1
python scripts/rocpd_timeline.py /tmp/kt/r_results.db 12 > $O/timeline_last_step.txt 2>&1
find /tmp/kt -name "*stats*" -name "*.csv" -exec cp {} $O/ \; 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
python scripts/rocpd_pmc.py /tmp/pf/r_results.db /tmp/pw/r_results.db $O/pmc_traffic.json > $O/pmc_hbm_traffic_per_kernel.txt 2>&1
ls -la $O
tail -1 $O/bench_default.log | cut -c1-300
