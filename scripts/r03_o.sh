#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_o}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv4_bwd" -s > $O/t_c4.log 2>&1; echo "conv4 tests rc $?"; grep "conv4 bwd\|passed\|failed\|Error" $O/t_c4.log | head -20
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['final_loss'])"; }
run fused
TUBER_NO_CONV4_BWD_FUSED=1 run unfused
run fused2
TUBER_NO_CONV4_BWD_FUSED=1 run unfused2
