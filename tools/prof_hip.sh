#!/bin/bash
# usage: tools/prof_hip.sh TAG : kernel trace + HIP API trace (host-side launch timestamps) of a short bench run
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $R/bench.py --no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --steps 6 --warmup 3 "$@" > $R/gpurun_out/${TAG}_bench.log 2>&1
ls -la $R/gpurun_out/prof_$TAG
