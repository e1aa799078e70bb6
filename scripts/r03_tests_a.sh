#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_b}
mkdir -p $O
timeout 900 python -m pytest tests/test_boundary_gpu.py -x -q -s -m gpu -k "reference_training_script" > $O/t_boundary.log 2>&1; echo "boundary rc $?"
timeout 900 python -m pytest tests/test_training_gpu.py -x -q -s -m gpu -k "world2 or graph_cache or one_rank or graph_replay" > $O/t_training.log 2>&1; echo "training rc $?"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -s -m gpu -k "teacher" > $O/t_teacher.log 2>&1; echo "teacher rc $?"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -s -m gpu -k "well_conditioned" > $O/t_deep.log 2>&1; echo "deep rc $?"
tail -3 $O/t_boundary.log $O/t_training.log $O/t_teacher.log $O/t_deep.log
