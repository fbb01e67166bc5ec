#!/bin/bash
# The same six training steps (same seeds, ordered reductions) with the kernel paths of rounds 3-4 on and off: the loss
# trajectories must agree to the arithmetic's noise (the first forward to ~4 digits, then at the rate the RoI sampling
# amplifies rounding).   tools/trajectory_ab.sh OUT
out=$1
F="--no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --no-other-bodies --steps 4 --warmup 2 --verbose"
CPM_DETERMINISTIC=1 CPM_RPN_SPARSE=2 python bench.py $F 2>/dev/null | grep "^warmup\|^step" > $out.new.txt
CPM_DETERMINISTIC=1 CPM_WGRAD_TAPS=0 CPM_WGRAD_KS=1 CPM_W4=0 CPM_STEM_FUSED=0 CPM_RPN_PRED_FUSED=0 CPM_CLEAR_GRADS_IN_STEP=0 \
  CPM_RPN_SPARSE=0 CPM_ROI_BWD_GROUP=0 CPM_CHAIN_FILL=0 CPM_FWD_SIDE=0 CPM_IGEMM_PT=0 CPM_BENCH_NCHW=1 \
  python bench.py $F 2>/dev/null | grep "^warmup\|^step" > $out.old.txt
paste -d'\n' $out.new.txt $out.old.txt
