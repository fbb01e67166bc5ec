#!/bin/bash
# usage (GPU box): tools/pmc_layer.sh TAG LAYERS  -> SQ occupancy / wait / LDS counters of the conv kernels of
# tools/bench_conv.py --filter LAYERS (bf16x3), one --pmc pass per counter group, summed per kernel name into
# gpurun_out/TAG_layer_pmc.txt.  Kernel trace only, as gpurun requires.
TAG=$1; LAYERS=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_lp$i -o p -- python3 $R/tools/bench_conv.py --math bf16x3 --filter $LAYERS --iters 3 > $R/gpurun_out/${TAG}_lp$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
cd $R
python3 - $TAG <<'PY'
import collections, csv, glob, re, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int); dur = collections.defaultdict(float)
first = True
for i in range(1, 8):
    for f in glob.glob("gpurun_out/%s_lp%d/**/*counter_collection.csv" % (tag, i), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = re.sub(r"\(.*", "", nm)[:50] + " g%s" % r["Grid_Size"]
            if not re.search(r"igemm|wgrad", k): continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if i == 1 and key not in seen:
                seen.add(key); cnt[k] += 1; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
with open("gpurun_out/%s_layer_pmc.txt" % tag, "w") as o:
    for k in sorted(agg):
        n = max(cnt[k], 1)
        o.write("%s  launches %d  avg %.1f us\n" % (k, cnt[k], dur[k] / n))
        for c in sorted(agg[k]):
            o.write("    %-32s %16.0f per launch\n" % (c, agg[k][c] / n))
print(open("gpurun_out/%s_layer_pmc.txt" % tag).read())
PY
