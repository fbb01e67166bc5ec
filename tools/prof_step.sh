#!/bin/bash
# usage: tools/prof_step.sh TAG [extra bench args]   (on the GPU box) -> gpurun_out/prof_TAG/, TAG_timeline.txt, TAG_gaps.txt
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $R/bench.py --no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --no-other-bodies --steps 10 --warmup 3 "$@" > $R/gpurun_out/${TAG}_bench.log 2>&1
cd $R
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv")
python tools/step_timeline.py $T > gpurun_out/${TAG}_timeline.txt
python tools/step_gaps.py $T > gpurun_out/${TAG}_gaps.txt
tail -1 gpurun_out/${TAG}_timeline.txt
head -3 gpurun_out/${TAG}_gaps.txt
grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench.log | head -1
