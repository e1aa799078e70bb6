#!/bin/bash
# round 4, first GPU call: full GPU suite on the refactored tree + default bench line
mkdir -p gpurun_out/r04_a
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r04_a/pytest.log
python bench.py --steps 30 > gpurun_out/r04_a/bench.json 2> gpurun_out/r04_a/bench.err
tail -c 3000 gpurun_out/r04_a/pytest.log
cat gpurun_out/r04_a/bench.json | cut -c1-1200
