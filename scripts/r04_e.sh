#!/bin/bash
mkdir -p gpurun_out/r04_e
python -m pytest tests/test_training_gpu.py tests/test_model_gpu.py -q -x 2>&1 | tail -15 > gpurun_out/r04_e/tests.log
python bench.py --steps 30 --no-cpu-baseline --no-roofline > gpurun_out/r04_e/bench_fork.json 2> gpurun_out/r04_e/bench_fork.err
TUBER_AB=no_class_branch_fork python bench.py --steps 30 --no-cpu-baseline --no-roofline > gpurun_out/r04_e/bench_nofork.json 2> gpurun_out/r04_e/bench_nofork.err
python bench.py --steps 30 --no-cpu-baseline --no-roofline > gpurun_out/r04_e/bench_fork2.json 2>> gpurun_out/r04_e/bench_fork.err
tail -8 gpurun_out/r04_e/tests.log; for f in fork nofork fork2; do cut -c1-330 gpurun_out/r04_e/bench_$f.json | grep -o '"ms_per_step": [0-9.]*'; done; tail -3 gpurun_out/r04_e/bench_fork.err
