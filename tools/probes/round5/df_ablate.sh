#!/bin/bash
mkdir -p gpurun_out/r5
for abl in 0 1 8 2 3 4 7 16 32 64 112; do
  for off in 1.5 0.3; do
    echo "== ABL=$abl off=$off"
    CPM_DF_ABL=$abl timeout -k 10 120 python tools/time_deform.py --only fused --offset-range $off 2>/dev/null | grep "C=1024\|C=512"
  done
done
