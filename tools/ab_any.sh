#!/bin/bash
# usage: tools/ab_any.sh "ENV_A" "ENV_B" [pairs] -- alternates two environments (e.g. "CPM_X=0" "CPM_X=1 CPM_Y=2") on the
# default training bench (headline only) and prints ms per step of every run and the two means
A=$1; B=$2; N=${3:-4}
F="--no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --no-other-bodies --steps 40 --warmup 10"
for i in $(seq $N); do
  for e in "$A" "$B"; do
    env $e python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$e', d['ms_per_step'])"
  done
done | tee /tmp/ab_any.txt
python - <<PY
import collections
acc = collections.defaultdict(list)
for l in open("/tmp/ab_any.txt"):
    k, v = l.rsplit(" ", 1)
    acc[k].append(float(v))
for k, v in acc.items():
    print("mean %-40s %.3f ms over %d runs (min %.2f max %.2f)" % (k, sum(v) / len(v), len(v), min(v), max(v)))
PY
