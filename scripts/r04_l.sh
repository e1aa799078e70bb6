#!/bin/bash
python -m pytest tests/test_model_gpu.py -q -x -k "deferred_weight or two_rank" 2>&1 | tail -3
python -m pytest tests/test_training_gpu.py -q -x -k "every_ab_switch" 2>&1 | tail -3
