mkdir -p gpurun_out/r05_fin2
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/r05_fin2/suite_s.log 2>&1; echo "suite rc $?" >> gpurun_out/r05_fin2/suite_s.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_fin2/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r05_fin2/smoke.log
QUICK=1 bash scripts/collect_r05_all.sh r05_fin2q > gpurun_out/r05_fin2/collect.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r05_fin2q/bench_ab_default.json 2>/dev/null
tail -4 gpurun_out/r05_fin2/suite_s.log; tail -2 gpurun_out/r05_fin2/smoke.log; head -2 gpurun_out/r05_fin2q/timeline_last_step.txt; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r05_fin2q/bench_default.log gpurun_out/r05_fin2q/bench_ab_default.json
