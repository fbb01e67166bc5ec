#!/bin/bash
# tile choice on the short-reduction 1x1 layers (8-64 k-steps), 30 iterations per point
for L in l3_1x1_256_1024 fpn_lat_p2 l4_1x1_512_2048 l4_1x1_2048_512 l3_1x1_1024_256 l2_1x1_512_128 l1_1x1_256_64; do
  for F in planner "64,64,1" "128,64,1" "128,128,1" planner "64,64,1"; do
    echo -n "$L $F : "
    if [ "$F" = planner ]; then python tools/bench_conv.py --math bf16x3 --filter $L --iters 30 2>&1 | grep "^$L" | awk '{print "fwd",$4,"dgrad",$7}'
    else CPM_IGEMM_FORCE=$F python tools/bench_conv.py --math bf16x3 --filter $L --iters 30 2>&1 | grep "^$L" | awk '{print "fwd",$4,"dgrad",$7}'; fi
  done
done
