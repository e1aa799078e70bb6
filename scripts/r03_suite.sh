#!/bin/bash
# the whole -m gpu suite + smoke + default bench on one box; logs under gpurun_out/$1
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r03_suite}
mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "pytest rc $?"
tail -n 4 $O/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
head -c 600 $O/bench_default.json
