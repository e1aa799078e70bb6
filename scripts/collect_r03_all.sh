#!/bin/bash
# round 3 evidence set -> gpurun_out/r03_z/: profile passes of the default command + the other BASELINE configs + the A/B lines DESIGN.md cites
bash scripts/collect_profiles_r02.sh r03_z
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_z
python scripts/rocpd_sequence.py /tmp/kt/r_results.db 0 40 > $O/sequence_last_step.txt 2>&1
B="timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
$B --pretrained-freeze > $O/bench_freeze.json 2>/dev/null
$B --config TubeR_CSN50_AVA21.yaml > $O/bench_cfg2_csn50_decode.json 2>/dev/null
$B --config Tuber_CSN152_JHMDB.yaml --height 288 --width 384 > $O/bench_cfg5_jhmdb.json 2>/dev/null
$B --eager --no-roofline > $O/bench_eager.json 2>/dev/null
$B --no-roofline --with-input-pipeline > $O/bench_with_input_pipeline.json 2>/dev/null
TUBER_FORCE_DDP=1 $B --no-roofline > $O/bench_force_ddp_one_rank.json 2>/dev/null
TUBER_FORCE_DDP=1 TUBER_NO_SPLIT_GRAPH=1 $B --no-roofline > $O/bench_force_ddp_one_rank_single_graph.json 2>/dev/null
TUBER_NO_BN3_IN_DW=1 $B --no-roofline > $O/bench_ab_no_bn3_in_dw.json 2>/dev/null
TUBER_TN_NO_BIG_TILES=1 $B --no-roofline > $O/bench_ab_no_big_tiles.json 2>/dev/null
TUBER_NO_ENTRY_CONV=1 TUBER_NO_PROJ_BWD_FUSED=1 TUBER_NO_STEM_BN_IN_WGRAD=1 $B --no-roofline > $O/bench_ab_no_first_block_and_stem_fusions.json 2>/dev/null
TUBER_NO_DW_BWD_ONE_LAUNCH=1 $B --no-roofline > $O/bench_ab_no_dw_bwd_one_launch.json 2>/dev/null
TUBER_NO_CONV4_BWD_FUSED=1 $B --no-roofline > $O/bench_ab_no_conv4_bwd_fused.json 2>/dev/null
TUBER_NO_BLOCKOUT_CONV1=1 $B --no-roofline > $O/bench_ab_no_blockout_conv1.json 2>/dev/null
TUBER_NO_CONV1_BWD_FUSED=1 $B --no-roofline > $O/bench_ab_no_conv1_bwd_fused.json 2>/dev/null
TUBER_NO_CONV4_BWD_FUSED=1 TUBER_NO_BLOCKOUT_CONV1=1 TUBER_NO_CONV1_BWD_FUSED=1 $B --no-roofline > $O/bench_ab_no_layer1_fused_kernels.json 2>/dev/null
$B --no-roofline > $O/bench_ab_default.json 2>/dev/null
for f in bench_freeze bench_cfg2_csn50_decode bench_cfg5_jhmdb bench_eager bench_with_input_pipeline bench_force_ddp_one_rank bench_force_ddp_one_rank_single_graph bench_ab_no_bn3_in_dw bench_ab_no_big_tiles bench_ab_no_first_block_and_stem_fusions bench_ab_no_dw_bwd_one_launch bench_ab_no_conv4_bwd_fused bench_ab_no_blockout_conv1 bench_ab_no_conv1_bwd_fused bench_ab_no_layer1_fused_kernels bench_ab_default; do
  python -c "import json; d=json.load(open('$O/$f.json')); print('%-44s %8.3f ms  %s %s' % ('$f', d['ms_per_step'], d.get('comm', ''), d.get('input_pipeline', '')))" 2>&1 | cut -c1-400
done
python scripts/gemm_bench.py tngroup > $O/microbench_tn_group.txt 2>&1
