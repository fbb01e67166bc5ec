#!/bin/bash
# grid-head 576x576x3x3 layers on 192-wide ring tiles: "cfg:split" pairs
out=$1; shift
for cs in "$@"; do
  c=${cs%%:*}; sp=${cs##*:}
  bm=$(echo $c | cut -d, -f1); bn=$(echo $c | cut -d, -f2)
  export CPM_RING_CFG=$c CPM_IGEMM_FORCE="$bm,$bn,$sp"
  echo "== cfg $c split $sp" >> ${out}.txt
  timeout -k 10 200 python tools/bench_conv.py --math sp --filter grid_conv_R 2>&1 | grep "grid_conv" | cut -c1-62 >> ${out}.txt || echo "FAILED" >> ${out}.txt
done
