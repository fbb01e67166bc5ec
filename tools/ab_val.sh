#!/bin/bash
# usage: tools/ab_val.sh VAR A B [bench args]: alternates VAR=A / VAR=B runs of the default training bench
VAR=$1; A=$2; B=$3; shift 3
F="--no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --steps 30 --warmup 8 $@"
for i in 1 2 3 4 5; do
for v in $A $B; do
env $VAR=$v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$VAR=$v', d['ms_per_step'], d['config']['roi_counts_last_step'])"
done
done
uptime
