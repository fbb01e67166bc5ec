set -e
F="--no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --steps 30 --warmup 8"
for i in 1 2; do
CPM_DEVICE_LISTS=0 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lists=0', d['ms_per_step'], d['config']['roi_counts_last_step'])"
CPM_DEVICE_LISTS=1 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lists=1', d['ms_per_step'], d['config']['roi_counts_last_step'])"
done
uptime
