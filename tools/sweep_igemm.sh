# per-layer sweep of the igemm tile / split-K choice (CPM_IGEMM_FORCE="bm,bn,split"); MATH=f32|bf16x3
MATH=${MATH:-bf16x3}
for L in l3_3x3_256 l4_3x3_512 l2_3x3_128 l3_1x1_256_1024 l3_1x1_1024_256 l4_1x1_512_2048 l4_1x1_2048_512 l2_1x1_128_512 l2_1x1_512_128 grid_conv_R64 l1_1x1_64_256 fpn_out_p4 fc7_R1024; do
  echo -n "$L planner : "; python tools/bench_conv.py --math $MATH --filter $L --iters 5 2>&1 | grep "^$L" | awk '{print "fwd",$5,$6,"dgrad",$8,$9}'
  for F in "64,64,1" "64,64,2" "64,64,4" "128,64,1" "128,64,2" "128,64,4" "128,128,1" "128,128,2" "128,128,4"; do
    echo -n "$L $F : "; CPM_IGEMM_FORCE=$F python tools/bench_conv.py --math $MATH --filter $L --iters 5 2>&1 | grep "^$L" | awk '{print "fwd",$5,$6,"dgrad",$8,$9}'
  done
done
