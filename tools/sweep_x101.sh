#!/bin/bash
# X-101 bs=1 dense 1x1s (1024 -> 1024 on 50x84 and its neighbours): the planner's choice against forced tile / split
# choices, persistent kernel on and off
mkdir -p gpurun_out
out=gpurun_out/sweep_x101.txt
: > $out
for pt in 2 0; do
  echo "== default plan, CPM_IGEMM_PT=$pt" >> $out
  CPM_IGEMM_PT=$pt CPM_IGEMM_DEBUG=1 python tools/bench_conv.py --filter x_l --math w4 --iters 20 --epi res >> $out 2>&1
done
for f in 128,128,1 128,128,2 128,64,1 128,64,2 64,64,1 64,64,2 128,32,1; do
  for pt in 2 0; do
    echo "== force $f CPM_IGEMM_PT=$pt" >> $out
    CPM_IGEMM_PT=$pt CPM_IGEMM_FORCE=$f python tools/bench_conv.py --filter x_l3_1x1 --math w4 --iters 20 --epi res >> $out 2>&1
  done
done
