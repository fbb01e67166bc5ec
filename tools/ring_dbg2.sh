#!/bin/bash
# timing-only ablation on grid_conv_R192 for a list of "cfg:split" pairs x CPM_RING_DBG modes
out=$1; shift
for cs in "$@"; do
  c=${cs%%:*}; sp=${cs##*:}
  bm=$(echo $c | cut -d, -f1); bn=$(echo $c | cut -d, -f2)
  export CPM_RING_CFG=$c CPM_IGEMM_FORCE="$bm,$bn,$sp"
  for d in 0 2 1 4 5; do
    export CPM_RING_DBG=$d
    echo "== cfg $c split $sp dbg $d" >> ${out}.txt
    timeout -k 10 200 python tools/bench_conv.py --math sp --filter grid_conv_R192 2>&1 | grep "grid_conv" | cut -c1-62 >> ${out}.txt || echo "FAILED" >> ${out}.txt
  done
done
