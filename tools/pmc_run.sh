#!/bin/bash
# usage (GPU box): tools/pmc_run.sh TAG  -> three separate --pmc passes of a short bench run (FETCH_SIZE, WRITE_SIZE,
# MFMA busy cycles + GRBM_GUI_ACTIVE), summarised into gpurun_out/TAG_pmc.json.  Kernel trace only, as gpurun requires.
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  CPM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_$N -o p -- python3 $R/bench.py --no-cpu-baseline --no-inference --no-roofline --no-other-math --no-full-rois --no-other-bodies --steps 3 --warmup 2 > $R/gpurun_out/${TAG}_$N.log 2>&1
  echo "pass $N done"
done
cd $R
python tools/pmc_summary.py gpurun_out/${TAG}_pmc.json FETCH=$(find gpurun_out/${TAG}_FETCH_SIZE -name "*counter_collection.csv") WRITE=$(find gpurun_out/${TAG}_WRITE_SIZE -name "*counter_collection.csv") MFMA=$(find gpurun_out/${TAG}_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv")
cat gpurun_out/${TAG}_pmc.json | head -60
