#!/bin/bash
# per-layer sweep of the weight gradient's pixel split (CPM_WGRAD_SPLIT), bf16x3
for L in grid_conv_R64 grid_conv_R192 l3_3x3_256 l4_3x3_512 fpn_out_p4 fc6_R1024 l3_1x1_256_1024 l2_3x3_128 grid_conv0_R64 iou_fc1_R64 fc7_R1024 l4_1x1_512_2048; do
  echo -n "$L planner : "; python tools/bench_conv.py --math bf16x3 --filter $L --iters 5 2>&1 | grep "^$L" | awk '{print "wgrad",$10,$11}'
  for F in 1 2 3 4 6 8 12 16 24; do
    echo -n "$L split $F : "; CPM_WGRAD_SPLIT=$F python tools/bench_conv.py --math bf16x3 --filter $L --iters 5 2>&1 | grep "^$L" | awk '{print "wgrad",$10,$11}'
  done
done
