#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_c}
mkdir -p $O
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_late.json 2> $O/bench_late.err
TUBER_NO_LATE_WGRAD=1 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_nolate.json 2> $O/bench_nolate.err
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_late2.json 2>> $O/bench_late.err
for f in bench_late bench_nolate bench_late2; do python -c "import json,sys; d=json.load(open('$O/$f.json')); print('$f', d['ms_per_step'], d['final_loss'])"; done
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -s -m gpu -k "well_conditioned or teacher" > $O/t_deep.log 2>&1; echo "deep rc $?"
timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_training_gpu.py -x -q -m gpu > $O/t_bt.log 2>&1; echo "boundary+training rc $?"
tail -n 3 $O/t_deep.log; tail -n 3 $O/t_bt.log
