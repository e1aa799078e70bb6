#!/bin/bash
mkdir -p gpurun_out/r04_suite
python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r04_suite/pytest.log
tail -5 gpurun_out/r04_suite/pytest.log
