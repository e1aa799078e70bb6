#!/bin/bash
# final validation of the round: full GPU suite, smoke, then the evidence set
mkdir -p gpurun_out/r04_suite
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r04_suite/pytest.log; tail -3 gpurun_out/r04_suite/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r04_suite/smoke.log
bash scripts/collect_r04_all.sh r04_z
