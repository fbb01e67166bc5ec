#!/bin/bash
# usage: tools/ab_envs.sh "VAR=a VAR=b ..." [bench args]: alternates the given environment settings on the default
# training bench (three rounds, one process per run, same box)
SETS=$1; shift
F="--no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --no-other-bodies --steps 30 --warmup 8 $@"
for i in 1 2 3; do
for v in $SETS; do
env $v python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['ms_per_step'])"
done
done
uptime
