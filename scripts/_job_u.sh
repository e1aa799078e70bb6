mkdir -p gpurun_out/r05_u
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm or in_projection" > gpurun_out/r05_u/kern.log 2>&1; echo "kern rc $?" >> gpurun_out/r05_u/kern.log
timeout 1200 python -m pytest tests/test_training_gpu.py -x -q -m gpu -s -k "no_ln_bwd_fusion or no_in_proj_dx2 or cooperative_decoder or graph" > gpurun_out/r05_u/train.log 2>&1; echo "train rc $?" >> gpurun_out/r05_u/train.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "backward_vs_oracle or spread or golden" > gpurun_out/r05_u/model.log 2>&1; echo "model rc $?" >> gpurun_out/r05_u/model.log
for i in 1 2; do
  TUBER_AB=no_ln_bwd_fusion,no_in_proj_dx2 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r05_u/bench_off.$i.log 2>&1
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r05_u/bench_on.$i.log 2>&1
done
tail -3 gpurun_out/r05_u/kern.log; tail -3 gpurun_out/r05_u/train.log; tail -3 gpurun_out/r05_u/model.log
for f in gpurun_out/r05_u/bench_*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"launches_per_step": [0-9]*' $f | head -1); done
