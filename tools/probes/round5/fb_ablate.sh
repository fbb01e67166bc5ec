#!/bin/bash
# timing-only ablations of igemm_fb_kernel<128,128> (CPM_FB_ABL bits: 1 no activation loads, 2 no split / LDS stores,
# 4 no LDS fragment reads, 8 no weight-fragment loads, 16 no barrier, 32 no MFMAs); results of the ablated runs are wrong
out=${1:-gpurun_out/r5/fb_abl.txt}
mkdir -p $(dirname $out)
: > $out
for abl in 0 1 2 3 4 6 7 8 9 11 15 16 23 31 32 47; do
  for force in "128,128,1" ""; do
    echo "== ABL=$abl FORCE=$force" >> $out
    CPM_IGEMM_FB=1 CPM_FB_ABL=$abl CPM_IGEMM_FORCE=$force timeout -k 10 120 python tools/bench_conv.py --math w4 --only fwd --iters 20 \
      --filter grid_conv_R34,grid_conv_R192,l3_1x1_1024_256,fpn_out_p3 2>&1 | grep -v "^layer\|weighted" | awk '{print $1, $4, $5}' >> $out
  done
done
