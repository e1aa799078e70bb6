#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_k}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bn_finalize" > $O/t_bn.log 2>&1; echo "bn tests rc $?"; tail -n 3 $O/t_bn.log
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['final_loss'], d.get('input_pipeline'))"; }
run default
TUBER_NO_BN_WIDE_FINALIZE=1 run no_wide_finalize
run default2
TUBER_NO_BN_WIDE_FINALIZE=1 run no_wide_finalize2
run with_pipeline --with-input-pipeline
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_training_gpu.py -x -q -m gpu > $O/t_mt.log 2>&1; echo "model+training rc $?"; tail -n 2 $O/t_mt.log
