#!/bin/bash
# Instruction mix and LDS bank conflicts of the GEMM / depthwise / attention kernels (PMC passes only: --kernel-trace + --pmc).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/mix
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -d /tmp/m1 -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/pass1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/m2 -o r -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/pass2.log 2>&1
python scripts/rocpd_instmix.py /tmp/m1/r_results.db /tmp/m2/r_results.db > $O/instmix.txt 2>&1
tail -3 $O/pass1.log | cut -c1-200; grep -i "attn\|kernel   " $O/instmix.txt
