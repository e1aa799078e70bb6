#!/bin/bash
mkdir -p gpurun_out/r04_c
python -m pytest tests/test_kernels_gpu.py -q -x -k "dwconv_tile_backward_with_bn" 2>&1 | tail -5 > gpurun_out/r04_c/kernel_test.log
python scripts/gemm_bench.py dwboth 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_c/dwboth.txt
python -m pytest tests/test_training_gpu.py -q -x -k "every_ab_switch" 2>&1 | tail -15 > gpurun_out/r04_c/ab_test.log
python bench.py --steps 30 --no-cpu-baseline > gpurun_out/r04_c/bench.json 2> gpurun_out/r04_c/bench.err
cat gpurun_out/r04_c/kernel_test.log gpurun_out/r04_c/dwboth.txt; tail -6 gpurun_out/r04_c/ab_test.log; cut -c1-400 gpurun_out/r04_c/bench.json
