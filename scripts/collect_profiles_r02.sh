#!/bin/bash
# Runs on the GPU box (via gpurun): default bench line, rocprofv3 kernel-trace stats of the same command, HBM traffic (FETCH_SIZE /
# WRITE_SIZE in separate passes), MFMA utilisation and instruction mix (PMC passes carry --kernel-trace only) -> gpurun_out/$1/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-prof}
ARGS="${@:2}"
mkdir -p $O
python bench.py $ARGS > $O/bench_default.log 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline $ARGS > $O/bench_under_rocprof.log 2>&1
python scripts/rocpd_summary.py /tmp/kt/r_results.db 13 > $O/kernel_trace_stats.txt 2>&1
python scripts/rocpd_timeline.py /tmp/kt/r_results.db 14 > $O/timeline_last_step.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $ARGS > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $ARGS > $O/pmc_write.log 2>&1
python scripts/rocpd_pmc.py /tmp/pf/r_results.db /tmp/pw/r_results.db $O/pmc_traffic.json > $O/pmc_hbm_traffic_per_kernel.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/pm -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $ARGS > $O/pmc_mfma.log 2>&1
python scripts/rocpd_mfma_util.py /tmp/pm/r_results.db $O/pmc_mfma_util.json > $O/pmc_mfma_util.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -d /tmp/m1 -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline $ARGS > $O/pmc_mix1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/m2 -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline $ARGS > $O/pmc_mix2.log 2>&1
python scripts/rocpd_instmix.py /tmp/m1/r_results.db /tmp/m2/r_results.db > $O/instmix.txt 2>&1
tail -1 $O/bench_default.log | cut -c1-400
head -30 $O/pmc_mfma_util.txt
