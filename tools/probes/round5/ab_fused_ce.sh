#!/bin/bash
# alternates two environments: tools-style A/B of a switch on the default training bench
F="--no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --no-other-bodies --steps 40 --warmup 10"
for i in 1 2 3 4; do
for v in 0 1; do
env CPM_FUSED_CE=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fused_ce=$v', d['ms_per_step'], d['config']['roi_counts_last_step'])"
done
done
