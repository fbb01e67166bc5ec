#!/bin/bash
# usage (GPU box): tools/refresh_profiles.sh TAG -> everything profiles/ is refreshed from, under gpurun_out/:
# kernel traces of the step (one stream / two streams), PMC passes (step and per layer), the per-shape conv times,
# the default bench line.  Each part logs to its own file; a failing part does not stop the rest.
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
CPM_WGRAD_STREAM=0 tools/prof_step.sh ${TAG}_one > gpurun_out/${TAG}_one.log 2>&1; echo "one stream: $(tail -1 gpurun_out/${TAG}_one.log)"
tools/prof_step.sh ${TAG}_two > gpurun_out/${TAG}_two.log 2>&1; echo "two streams: $(tail -1 gpurun_out/${TAG}_two.log)"
tools/pmc_run.sh ${TAG} > gpurun_out/${TAG}_pmc.log 2>&1; echo "pmc done"
tools/pmc_layer.sh ${TAG} "fpn_out_p2,grid_conv_R64,l3_1x1_256_1024" > gpurun_out/${TAG}_layer.log 2>&1; echo "layer pmc done"
CPM_PROF_DUMP=$R/gpurun_out/${TAG}_shapes.csv python bench.py --no-cpu-baseline --no-inference --no-other-math --no-full-rois --no-other-bodies > gpurun_out/${TAG}_shapes.log 2>&1
python tools/dump_conv_shapes.py gpurun_out/${TAG}_shapes.csv > gpurun_out/${TAG}_conv_shapes_default.txt 2>&1
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_err.log
tail -c 600 gpurun_out/${TAG}_bench_line.json
