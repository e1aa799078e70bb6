#!/bin/bash
# round 6: the one-rank forced-DDP line: hybrid edges (counter at the first cut, events behind it; default) / all counters / all events / no edges / plain.  Same box, two rounds.
O=gpurun_out/${1:-r06_ddp_hybrid}
mkdir -p $O
B="timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline"
for r in 1 2; do
  $B > $O/plain_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 $B > $O/ddp_cut43_hybrid_$r.json 2>$O/err_hybrid.txt
  TUBER_FORCE_DDP=1 TUBER_DDP_EDGE=event $B > $O/ddp_cut43_event_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_DDP_EDGE=flag $B > $O/ddp_cut43_flag_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_DDP_CUTS=3 $B > $O/ddp_cut3_hybrid_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_DDP_CUTS=3 TUBER_DDP_EDGE=event $B > $O/ddp_cut3_event_round5_form_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_DDP_BF16=1 $B > $O/ddp_cut43_hybrid_bf16_$r.json 2>/dev/null
  TUBER_FORCE_DDP=1 TUBER_NO_OWN_RCCL=1 $B > $O/ddp_structure_only_$r.json 2>/dev/null
done
python scripts/r06_summ.py $O | tee $O/summary.txt
