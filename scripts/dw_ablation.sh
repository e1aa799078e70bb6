#!/bin/bash
# timing ablations of dwconv_tile_bwd_both_kernel (-DDW_DBG=1 no tap loop, 2 no park, 3 no output stores; results are wrong, times are not)
cd "$GRAFT_REPO_ROOT"
L=tubelet_transformer_amd/lib
cp $L/libtuber_hip.so /tmp/libtuber_hip.orig.so
python scripts/gemm_bench.py dwboth 2>&1 | grep "dw bwd" | sed 's/^/full     /'
for n in ${DBGS:-1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -DDW_DBG=$n -x hip -c tubelet_transformer_amd/csrc/dwconv_tile.hip -o /tmp/dwt_dbg.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libtuber_hip.so $(ls $L/*.o | grep -v dwconv_tile) /tmp/dwt_dbg.o
  python scripts/gemm_bench.py dwboth 2>&1 | grep "dw bwd" | sed "s/^/DW_DBG=$n /"
done
cp /tmp/libtuber_hip.orig.so $L/libtuber_hip.so
