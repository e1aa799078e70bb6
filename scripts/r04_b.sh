#!/bin/bash
mkdir -p gpurun_out/r04_b
python -m pytest tests/test_kernels_gpu.py -q -x -k "dwconv_tile_backward_with_bn" 2>&1 | tail -25 > gpurun_out/r04_b/kernel_test.log
python scripts/gemm_bench.py dwboth > gpurun_out/r04_b/dwboth.txt 2>&1
python -m pytest tests/test_training_gpu.py -q -x -k "every_ab_switch" 2>&1 | tail -25 > gpurun_out/r04_b/ab_test.log
python bench.py --steps 30 --no-cpu-baseline > gpurun_out/r04_b/bench.json 2> gpurun_out/r04_b/bench.err
cat gpurun_out/r04_b/kernel_test.log gpurun_out/r04_b/dwboth.txt; tail -12 gpurun_out/r04_b/ab_test.log; cut -c1-400 gpurun_out/r04_b/bench.json
