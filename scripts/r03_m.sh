#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_m}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "dwconv or folded" > $O/t_dw.log 2>&1; echo "dw tests rc $?"; tail -n 3 $O/t_dw.log
TUBER_DW_CQ=8 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "dwconv or folded" > $O/t_dw8.log 2>&1; echo "dw tests (cq 8 forced) rc $?"; tail -n 2 $O/t_dw8.log
for cq in 16 8; do echo "CQ $cq"; TUBER_DW_CQ=$cq python scripts/gemm_bench.py dw 2>&1 | grep -v amdgpu.ids; done
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['final_loss'])"; }
run auto
TUBER_DW_CQ=16 run cq16
TUBER_DW_CQ=8 run cq8
run auto2
TUBER_DW_CQ=16 run cq16_2
