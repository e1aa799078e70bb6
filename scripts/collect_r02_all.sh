bash scripts/collect_profiles_r02.sh r02_z
O=gpurun_out/r02_z
timeout 200 python bench.py --steps 20 --warmup 3 --pretrained-freeze --no-cpu-baseline > $O/bench_freeze.json 2>/dev/null
timeout 200 python bench.py --steps 20 --warmup 3 --config TubeR_CSN50_AVA21.yaml --no-cpu-baseline > $O/bench_cfg2_csn50_decode.json 2>/dev/null
timeout 200 python bench.py --steps 20 --warmup 3 --config Tuber_CSN152_JHMDB.yaml --height 288 --width 384 --no-cpu-baseline > $O/bench_cfg5_jhmdb.json 2>/dev/null
for f in bench_freeze bench_cfg2_csn50_decode bench_cfg5_jhmdb; do head -c 300 $O/$f.json; echo; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --config Tuber_CSN152_JHMDB.yaml --height 288 --width 384 > $O/cfg5_under_rocprof.log 2>&1
python scripts/rocpd_summary.py /tmp/kt5/r_results.db 13 > $O/cfg5_kernel_trace_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --config TubeR_CSN50_AVA21.yaml > $O/cfg2_under_rocprof.log 2>&1
python scripts/rocpd_summary.py /tmp/kt2/r_results.db 13 > $O/cfg2_kernel_trace_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/ktf -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --pretrained-freeze > $O/freeze_under_rocprof.log 2>&1
python scripts/rocpd_summary.py /tmp/ktf/r_results.db 13 > $O/freeze_kernel_trace_stats.txt 2>&1
