#!/bin/bash
# usage: tools/ab_env.sh VAR [bench args]: alternates VAR=0 / VAR=1 runs of the default training bench
VAR=$1; shift
F="--no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --steps 30 --warmup 8 $@"
for i in 1 2 3; do
for v in 0 1; do
env $VAR=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$VAR=$v', d['ms_per_step'], d['config']['roi_counts_last_step'])"
done
done
uptime
