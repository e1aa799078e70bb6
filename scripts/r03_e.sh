#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_e}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm_tn" > $O/t_tn.log 2>&1; echo "tn tests rc $?"; tail -n 3 $O/t_tn.log
python scripts/gemm_bench.py tngroup > $O/tngroup_big.txt 2>&1
TUBER_TN_NO_BIG_TILES=1 python scripts/gemm_bench.py tngroup > $O/tngroup_small.txt 2>&1
echo BIG; cat $O/tngroup_big.txt | grep -v amdgpu.ids; echo SMALL; cat $O/tngroup_small.txt | grep -v amdgpu.ids
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_big.json 2> $O/bench_big.err
TUBER_TN_NO_BIG_TILES=1 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_small.json 2> $O/bench_small.err
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_big2.json 2>> $O/bench_big.err
for f in bench_big bench_small bench_big2; do python -c "import json,sys; d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['final_loss'])"; done
