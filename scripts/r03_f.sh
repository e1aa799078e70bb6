#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_f}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_tn or dwconv or folded" > $O/t_k.log 2>&1; echo "kernel tests rc $?"; tail -n 3 $O/t_k.log
for tg in 64 128 256; do echo "BIG target $tg"; TUBER_TN_BIG_TARGET=$tg python scripts/gemm_bench.py tngroup 2>&1 | grep -v amdgpu.ids | head -2; done
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['final_loss'])"; }
run default
TUBER_NO_BN3_IN_DW=1 run no_bn3_in_dw
TUBER_TN_NO_BIG_TILES=1 run no_big_tiles
TUBER_NO_BN3_IN_DW=1 TUBER_TN_NO_BIG_TILES=1 run neither
run default2
