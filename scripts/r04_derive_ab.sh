#!/bin/bash
# same-box A/B of the batched coefficient-derivation loads (bn_bwd_fa, dwconv_tile_bwd_both): new build first, then the previous sources
set -u
out=gpurun_out/r04_derive; mkdir -p $out
timeout 300 python scripts/gemm_bench.py bnfa 2>&1 | grep -v amdgpu.ids | tee $out/bnfa_new.txt
timeout 300 python scripts/gemm_bench.py dwboth 2>&1 | grep -v amdgpu.ids | tee $out/dwboth_new.txt
timeout 600 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/bench_new.json; python -c "import json;print('new', json.load(open('$out/bench_new.json'))['ms_per_step'])"
if [ -d tmp_old ]; then
  cp tmp_old/norm.hip tmp_old/dwconv_tile.hip tubelet_transformer_amd/csrc/
  python -c "from tubelet_transformer_amd import build; build.build(verbose=False)" 2>&1 | tail -2
  timeout 300 python scripts/gemm_bench.py bnfa 2>&1 | grep -v amdgpu.ids | tee $out/bnfa_old.txt
  timeout 300 python scripts/gemm_bench.py dwboth 2>&1 | grep -v amdgpu.ids | tee $out/dwboth_old.txt
  timeout 600 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/bench_old.json; python -c "import json;print('old', json.load(open('$out/bench_old.json'))['ms_per_step'])"
fi
