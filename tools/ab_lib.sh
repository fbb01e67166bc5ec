#!/bin/bash
# usage (GPU box): tools/ab_lib.sh BASE.so [bench args]: alternates the training bench between a saved build of the
# library (CPM_LIB=BASE.so) and the in-tree one
BASE=$1; shift
F="--no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --steps 30 --warmup 8 $@"
for i in 1 2 3; do
for v in base new; do
if [ $v = base ]; then export CPM_LIB=$BASE; else unset CPM_LIB; fi
python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['ms_per_step'], d['config']['roi_counts_last_step'])"
done
done
uptime
