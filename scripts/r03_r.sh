#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_r}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv1_bwd or conv4_bwd or blockout" -s > $O/t_f.log 2>&1; echo "fused kernel tests rc $?"; grep "conv1 bwd\|passed\|failed\|Error\|differs" $O/t_f.log | head -40
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "deferred_weight or golden" > $O/t_model.log 2>&1; echo "model tests rc $?"; tail -n 2 $O/t_model.log
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['final_loss'])"; }
run fused
TUBER_NO_CONV1_BWD_FUSED=1 run no_conv1_fused
run fused2
TUBER_NO_CONV1_BWD_FUSED=1 run no_conv1_fused2
