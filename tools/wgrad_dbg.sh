#!/bin/bash
# Timing-only ablations of wgrad_split_kernel<128,128> (CPM_WGRAD_DBG, see the kernel): prints the weight-gradient column
# of tools/bench_conv.py for a few layers per mode.   tools/wgrad_dbg.sh OUT [layer filter]
out=$1; flt=${2:-fpn_out_p2,fpn_out_p4,l3_1x1_256_1024,l4_3x3_512,grid_conv_R105,grid_conv_R34}
: > $out.txt
for ns in 0 1; do for d in 0 1 2 3; do
  echo "== dbg $d nostore $ns" >> $out.txt
  CPM_WGRAD_DBG=$d CPM_WGRAD_NOSTORE=$ns timeout -k 10 120 python tools/bench_conv.py --math bf16x3 --iters 20 --filter $flt 2>/dev/null \
    | awk 'NR>1 && $1 !~ /weighted/ {printf "%-20s wgrad %8s us %7s TF\n", $1, $10, $11}' >> $out.txt || exit 1
done; done
