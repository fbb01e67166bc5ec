#!/bin/bash
# timing-only ablation of the LDS-DMA ring kernel: CPM_RING_DBG bits 1 = no DMA, 2 = zero-record descriptors (the DMA
# instructions issue, nothing is fetched), 4 = no operand reads.  Results are WRONG in these builds; only times count.
out=$1; cfgs=$2; filt=${3:-fpn_out_p2,grid_conv_R192,l3_3x3}
for c in $cfgs; do
  export CPM_RING_CFG=$c
  for d in 0 2 1 4 5; do
    export CPM_RING_DBG=$d
    echo "== cfg $c dbg $d" >> ${out}.txt
    timeout -k 10 200 python tools/bench_conv.py --math sp --filter "$filt" 2>&1 | grep -v amdgpu.ids | cut -c1-62 >> ${out}.txt || echo "FAILED" >> ${out}.txt
  done
done
