#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -q -x -k "join" 2>&1 | tail -4
python -m pytest tests/test_training_gpu.py -q -x -k "every_ab_switch" 2>&1 | tail -3
python -m pytest tests/test_fullsize_gpu.py -q -x -k "teacher_forced or shallow_body" 2>&1 | tail -3
python bench.py --steps 30 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
TUBER_AB=no_strided_join_fusion,no_ds_join_fusion python bench.py --steps 30 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
python bench.py --steps 30 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
