#!/bin/bash
# Timing-only ablations of igemm_kernel (CPM_IGEMM_DBG: 8 nothing fetched, 16 no epilogue, 24 both): forward and data
# gradient columns of tools/bench_conv.py.   tools/igemm_dbg.sh OUT [layer filter]
out=$1; flt=${2:-l1_1x1_64_256,l1_1x1_256_64,l2_1x1_128_512,l3_1x1_256_1024,l3_1x1_1024_256,l3_3x3_256,l4_1x1_512_2048,fpn_lat_p2,grid_conv_R34,grid_conv_R105}
: > $out.txt
for d in 0 8 16 24; do
  echo "== dbg $d" >> $out.txt
  CPM_IGEMM_DBG=$d CPM_IGEMM_DEBUG=1 timeout -k 10 120 python tools/bench_conv.py --math bf16x3 --iters 20 --filter $flt 2>$out.plan \
    | awk 'NR>1 && $1 !~ /weighted/ {printf "%-20s fwd %8s us %7s TF | dgrad %8s us %7s TF\n", $1, $4, $5, $7, $8}' >> $out.txt || exit 1
done
sort -u $out.plan | grep "igemm plan" >> $out.txt
