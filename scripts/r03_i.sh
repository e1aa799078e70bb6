#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d.get('comm', {}).get('exposed_ms'))"; }
run single
TUBER_FORCE_SPLIT_GRAPH=1 run single_forced_split
TUBER_FORCE_DDP=1 TUBER_NO_OWN_RCCL=1 run ddp_no_transport
TUBER_FORCE_DDP=1 run ddp_own_rccl
TUBER_FORCE_DDP=1 TUBER_RCCL_IN_GRAPH=1 run ddp_in_graph
run single2
