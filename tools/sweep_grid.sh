#!/bin/bash
# grid-head 3x3 (576 -> 576 on 7x7 RoI maps): the forward / data-gradient kernel under forced tile + split choices
mkdir -p gpurun_out/r4
out=gpurun_out/r4/sweep_grid.txt
: > $out
echo "== default plan" >> $out
CPM_IGEMM_DEBUG=1 python tools/bench_conv.py --filter grid_conv_R --math w4 --iters 20 >> $out 2>&1
for f in 128,128,1 128,128,2 128,128,3 128,64,1 128,64,2 128,64,3 64,64,1 64,64,2 64,64,3; do
  echo "== force $f" >> $out
  CPM_IGEMM_FORCE=$f python tools/bench_conv.py --filter grid_conv_R --math w4 --iters 20 --only fwd >> $out 2>&1
done
