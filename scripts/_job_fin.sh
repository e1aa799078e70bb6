mkdir -p gpurun_out/r05_fin
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/r05_fin/suite_s.log 2>&1; echo "suite rc $?" >> gpurun_out/r05_fin/suite_s.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_fin/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r05_fin/smoke.log
QUICK=1 bash scripts/collect_r05_all.sh r05_finq > gpurun_out/r05_fin/collect.log 2>&1
for sw in no_ln_bwd_fusion no_in_proj_dx2 no_ln_bwd_fusion,no_in_proj_dx2 no_decoder_coop; do
  TUBER_AB=$sw timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r05_finq/bench_ab_${sw//,/+}.json 2>/dev/null
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r05_finq/bench_ab_default.json 2>/dev/null
tail -4 gpurun_out/r05_fin/suite_s.log; tail -2 gpurun_out/r05_fin/smoke.log; ls gpurun_out/r05_finq | head -40
