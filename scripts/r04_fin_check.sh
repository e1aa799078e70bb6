#!/bin/bash
# BatchNorm finalisation folded into the consumer kernels (bn1 -> depthwise forward, bn4 -> block output): kernel tests, A/B switch test,
# model tests, same-box step A/B
set -u
out=gpurun_out/r04_fin; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "dwconv_tile or block_out" 2>&1 | tail -5 | tee $out/kernels.log
timeout 1200 python -m pytest tests/test_training_gpu.py -x -q -m gpu -k "ab_switch" 2>&1 | tail -5 | tee $out/ab.log
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $out/model.log
for ab in "" ${AB:-no_bn1_in_dw_fwd} "" ${AB:-no_bn1_in_dw_fwd}; do
  TUBER_AB=$ab timeout 600 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/b.json
  python -c "import json;print('ab=[$ab]', json.load(open('$out/b.json'))['ms_per_step'])" | tee -a $out/step_ab.txt
done
