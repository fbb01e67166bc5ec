#!/bin/bash
# grid-head 3x3 (576 -> 576 on 7x7 RoI maps) at 34 / 64 / 105 RoIs: deeper reduction splits than tools/sweep_grid.sh tried
mkdir -p gpurun_out
out=gpurun_out/sweep_grid2.txt
: > $out
echo "== default plan" >> $out
CPM_IGEMM_DEBUG=1 python tools/bench_conv.py --filter grid_conv_R --math w4 --iters 20 --only fwd 2>&1 | grep -v "amdgpu.ids" | sort | uniq >> $out
for f in 128,128,4 128,128,6 128,128,8 128,64,4 128,64,6 128,64,8 64,64,4 64,64,8; do
  echo "== force $f" >> $out
  CPM_IGEMM_FORCE=$f python tools/bench_conv.py --filter grid_conv_R --math w4 --iters 20 --only fwd 2>&1 | grep "grid_conv" >> $out
done
