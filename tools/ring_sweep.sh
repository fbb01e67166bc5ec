#!/bin/bash
# A/B of LDS-DMA ring configurations on the per-layer conv benchmark (one process per configuration; same box).
#   tools/ring_sweep.sh OUT_PREFIX "cfg1 cfg2 ..." [filter]      cfg = bm,bn,waves,stages,pipe  or  "default"
out=$1; cfgs=$2; filt=${3:-fpn_out,grid_conv,l3_,l2_3x3,l4_3x3,fc6}
for c in $cfgs; do
  tag=$(echo $c | tr ',' '_')
  if [ "$c" = default ]; then unset CPM_RING_CFG; else export CPM_RING_CFG=$c; fi
  echo "== $c" >> ${out}.txt
  timeout -k 10 300 python tools/bench_conv.py --math sp --filter "$filt" >> ${out}.txt 2>&1 || echo "FAILED $c" >> ${out}.txt
done
