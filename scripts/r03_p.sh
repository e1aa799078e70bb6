#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_p}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "blockout_conv1 or conv4_bwd" -s > $O/t_f.log 2>&1; echo "fused kernel tests rc $?"; grep "fused\|passed\|failed\|Error" $O/t_f.log | head -30
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "golden or eval" > $O/t_model.log 2>&1; echo "model tests rc $?"; tail -n 2 $O/t_model.log
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['final_loss'])"; }
run fused
TUBER_NO_BLOCKOUT_CONV1=1 run no_blockout_conv1
run fused2
TUBER_NO_BLOCKOUT_CONV1=1 run no_blockout_conv1_2
