#!/bin/bash
mkdir -p gpurun_out/r04_h
python -m pytest tests/test_fullsize_gpu.py -q -x -s -k "eval_forward" 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r04_h/fullsize_eval.log
cat gpurun_out/r04_h/fullsize_eval.log | cut -c1-600
