#!/bin/bash
# where do the +0.5 ms of the one-rank forced-DDP line come from, if the one-rank in-place ncclAllReduce is a no-op on the device (stream busy 5 us)?
O=gpurun_out/${1:-r06_ddp_probe}
mkdir -p $O
B="timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline"
for r in 1 2; do
  $B > $O/plain_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 $B > $O/ddp_cut43_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_DDP_SKIP_CALL=1 $B > $O/ddp_cut43_skip_call_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_NO_SPLIT_GRAPH=1 TUBER_DDP_SKIP_CALL=1 $B > $O/ddp_single_graph_skip_call_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_NO_OWN_RCCL=1 $B > $O/ddp_structure_only_$r.json 2>/dev/null
done
for f in $O/*.json; do
  python -c "import json,sys; d=json.loads([l for l in open('$f').read().splitlines() if l.startswith('{')][-1]); c=d.get('comm') or {}; print('%-40s %8.3f ms  exposed %s  %s' % ('$(basename $f .json)', d['ms_per_step'], c.get('exposed_ms'), c.get('issue_points')))" 2>&1 | cut -c1-700
done | tee $O/summary.txt
