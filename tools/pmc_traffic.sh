#!/bin/bash
# usage (GPU box): tools/pmc_traffic.sh TAG "LAYER1 LAYER2 ..." -> memory-side traffic of the conv kernels per layer and
# direction against the layer's algorithmic bytes: separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md, HBM) of
# tools/bench_conv.py --math w4 --epi res --filter LAYER --only DIR; summary gpurun_out/TAG_traffic.txt.
TAG=$1; LAYERS=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in $LAYERS; do
  for D in fwd dgrad wgrad; do
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_tr/${L}_${D}_$C -o p -- python3 $R/tools/bench_conv.py --math w4 --epi res --filter $L --only $D --iters 4 > $R/gpurun_out/${TAG}_tr_${L}_${D}_$C.log 2>&1 || echo "$L $D $C failed"
    done
  done
  echo "$L done"
done
cd $R
python3 tools/pmc_traffic_summary.py $TAG "$LAYERS" | tee gpurun_out/${TAG}_traffic.txt
