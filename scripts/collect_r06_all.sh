#!/bin/bash
# round 6 evidence set -> gpurun_out/$1/ (default r06_z): profile passes of the default command + the other BASELINE configs + the A/B lines DESIGN.md cites
T=${1:-r06_z}
bash scripts/collect_profiles_r02.sh $T
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$T
python scripts/rocpd_sequence.py /tmp/kt/r_results.db 0 40 > $O/sequence_last_step.txt 2>&1
[ -n "$QUICK" ] && exit 0
B="timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
$B --pretrained-freeze > $O/bench_freeze.json 2>/dev/null
$B --config TubeR_CSN50_AVA21.yaml > $O/bench_cfg2_csn50_decode.json 2>/dev/null
$B --config Tuber_CSN152_JHMDB.yaml --height 288 --width 384 > $O/bench_cfg5_jhmdb.json 2>/dev/null
$B --eager --no-roofline > $O/bench_eager.json 2>/dev/null
$B --no-roofline --with-input-pipeline > $O/bench_with_input_pipeline.json 2>/dev/null
TUBER_FORCE_DDP=1 $B --no-roofline > $O/bench_force_ddp_one_rank.json 2>/dev/null
TUBER_FORCE_DDP=1 TUBER_DDP_CUTS=3 TUBER_DDP_EDGE=event $B --no-roofline > $O/bench_force_ddp_one_rank_round5_form_one_cut_event_edges.json 2>/dev/null
TUBER_FORCE_DDP=1 TUBER_NO_SPLIT_GRAPH=1 $B --no-roofline > $O/bench_force_ddp_one_rank_single_graph.json 2>/dev/null
TUBER_FORCE_DDP=1 TUBER_NO_OWN_RCCL=1 $B --no-roofline > $O/bench_force_ddp_structure_only_no_edges.json 2>/dev/null
TUBER_SHARE_GPU=1 TUBER_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $O/bench_two_ranks_one_gpu_gloo_self_spawned.json 2>/dev/null
for sw in no_fresh_reduce no_join_mask no_ln_bwd_fusion no_in_proj_dx2 no_decoder_coop no_bn3_in_dw no_dw_bwd_one_launch no_conv4_bwd_fused no_blockout_conv1 no_conv1_bwd_fused no_join_fusion no_wgrad_groups no_bn_bwd_fa no_bn1_in_dw_fwd no_strided_join_fusion,no_ds_join_fusion; do
  TUBER_AB=$sw $B --no-roofline > $O/bench_ab_${sw//,/+}.json 2>/dev/null
done
TUBER_AB=no_entry_conv,no_proj_bwd_fused,no_stem_bn_in_wgrad $B --no-roofline > $O/bench_ab_no_first_block_and_stem_fusions.json 2>/dev/null
TUBER_AB=no_conv4_bwd_fused,no_blockout_conv1,no_conv1_bwd_fused $B --no-roofline > $O/bench_ab_no_layer1_fused_kernels.json 2>/dev/null
TUBER_NT_WSK96=0 $B --no-roofline > $O/bench_ab_wsk_64_row_tiles_only.json 2>/dev/null
$B --no-roofline > $O/bench_ab_default.json 2>/dev/null
for f in $O/bench_*.json; do
  python -c "import json,sys; d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1]); print('%-72s %8.3f ms  %s %s' % ('$(basename $f .json)', d['ms_per_step'], d.get('comm', ''), d.get('input_pipeline', '')))" 2>&1 | cut -c1-700
done
python scripts/eval_latency.py 2>&1 | grep -v amdgpu.ids > $O/eval_latency_cfg3.txt
python scripts/gemm_bench.py dwboth 2>&1 | grep -v amdgpu.ids > $O/microbench_dw_both.txt
python scripts/gemm_bench.py tngroup 2>&1 | grep -v amdgpu.ids > $O/microbench_tn_group.txt
