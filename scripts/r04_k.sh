#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_k
python -m pytest tests/test_kernels_gpu.py -q -x -k "conv1_bwd_fused" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r04_k/bench.log 2>&1
python scripts/rocpd_summary.py /tmp/kt/r_results.db 8 > gpurun_out/r04_k/kernel_trace_stats.txt 2>&1
grep -i "conv1_bwd\|block_out_bwd\|calls" gpurun_out/r04_k/kernel_trace_stats.txt | cut -c1-200
python bench.py --steps 30 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
python bench.py --steps 30 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
