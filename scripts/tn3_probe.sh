#!/bin/bash
for v in 1 2 3; do
  touch tubelet_transformer_amd/csrc/gemm.hip
  TN3_DBG=$v python -c "from tubelet_transformer_amd import build; build.build(verbose=False)" 2>&1 | tail -1
  echo "=== TN3_DBG=$v"; timeout 300 python scripts/tn3_probe.py 2>&1 | grep -v amdgpu.ids
done
