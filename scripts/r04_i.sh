#!/bin/bash
mkdir -p gpurun_out/r04_i
python -m pytest tests/test_kernels_gpu.py -q -x -k "dwconv" 2>&1 | tail -3
python scripts/gemm_bench.py dw 2>&1 | grep -v amdgpu.ids
python scripts/gemm_bench.py dwboth 2>&1 | grep -v amdgpu.ids
python bench.py --steps 30 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
python bench.py --steps 30 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
