#!/bin/bash
# the +0.7 ms of the one-rank DDP line sit in the two stream edges around the (empty) exchange, not in RCCL: which event flavour removes them?
O=gpurun_out/${1:-r06_ddp_edge}
mkdir -p $O
B="timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline"
for r in 1 2; do
  $B > $O/plain_$r.json 2>/dev/null
  for e in torch plain device nofence; do
    TUBER_FORCE_DDP=1 TUBER_DDP_EDGE=$e $B > $O/ddp_cut43_edge_${e}_$r.json 2>$O/err_$e.txt
  done
  TUBER_FORCE_DDP=1 TUBER_DDP_EDGE=device TUBER_DDP_CUTS=3 $B > $O/ddp_cut3_edge_device_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_DDP_EDGE=device TUBER_NO_SPLIT_GRAPH=1 $B > $O/ddp_single_graph_edge_device_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_NO_OWN_RCCL=1 $B > $O/ddp_structure_only_$r.json 2>/dev/null
done
for f in $O/*.json; do
  python -c "import json,sys; d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1]); c=d.get('comm') or {}; print('%-40s %8.3f ms  exposed %s  %s' % ('$(basename $f .json)', d['ms_per_step'], c.get('exposed_ms'), [(p['issued_at_ms'], p['stream_busy_ms']) for p in (c.get('issue_points') or [])]))" 2>&1 | cut -c1-400
done | tee $O/summary.txt
