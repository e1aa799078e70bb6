#!/bin/bash
# round 6, VERDICT r05 item 1: what one rank of the DDP path costs with nothing on the wire -- plain step vs the forced one-rank
# communicator with the round-5 single cut (TUBER_DDP_CUTS=3) vs the two cuts (4,3), RCCL channel limits, bf16 windows.  Same box,
# two interleaved rounds.
O=gpurun_out/${1:-r06_ddp}
mkdir -p $O
B="timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline"
for r in 1 2; do
  $B > $O/plain_$r.json 2>$O/plain_$r.err
  TUBER_FORCE_DDP=1 TUBER_DDP_CUTS=3 $B > $O/ddp_cut3_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 $B > $O/ddp_cut43_$r.json 2>$O/ddp_cut43_$r.err
  TUBER_FORCE_DDP=1 TUBER_DDP_CUTS=4,3,2 $B > $O/ddp_cut432_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 NCCL_MAX_NCHANNELS=4 NCCL_MIN_NCHANNELS=4 $B > $O/ddp_cut43_ch4_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 NCCL_MAX_NCHANNELS=8 NCCL_MIN_NCHANNELS=8 $B > $O/ddp_cut43_ch8_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_DDP_BF16=1 $B > $O/ddp_cut43_bf16_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_NO_SPLIT_GRAPH=1 $B > $O/ddp_single_graph_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_NO_OWN_RCCL=1 $B > $O/ddp_structure_only_cut43_$r.json 2>/dev/null
done
for f in $O/*.json; do
  python -c "import json,sys; d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1]); c=d.get('comm') or {}; print('%-40s %8.3f ms  exposed %s  %s' % ('$(basename $f .json)', d['ms_per_step'], c.get('exposed_ms'), c.get('issue_points')))" 2>&1 | cut -c1-600
done | tee $O/summary.txt
