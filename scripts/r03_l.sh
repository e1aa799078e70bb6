#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_l}
mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; head -c 300 $O/bench_default.json; echo
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --with-input-pipeline > $O/bench_with_input_pipeline.json 2>/dev/null; tail -1 $O/bench_with_input_pipeline.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['input_pipeline'])"
