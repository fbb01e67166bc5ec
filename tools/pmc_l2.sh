#!/bin/bash
# usage (GPU box): tools/pmc_l2.sh TAG LAYERS -> L2 (TCC) request counters of the conv kernels of tools/bench_conv.py
TAG=$1; LAYERS=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_l2p$i -o p -- python3 $R/tools/bench_conv.py --math bf16x3 --filter $LAYERS --iters 3 > $R/gpurun_out/${TAG}_l2p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python3 - $TAG <<'PY'
import collections, csv, glob, re, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int); dur = collections.defaultdict(float)
for i in range(1, 7):
    for f in glob.glob("gpurun_out/%s_l2p%d/**/*counter_collection.csv" % (tag, i), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = re.sub(r"\(.*", "", nm)[:50] + " g%s" % r["Grid_Size"]
            if not re.search(r"igemm|wgrad", k) or "false>" in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"], i)
            if key not in seen:
                seen.add(key)
                if r["Counter_Name"] in ("TCC_REQ_sum", "TCC_HIT_sum", "TCP_TCC_READ_REQ_sum", "TCC_EA0_RDREQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"):
                    pass
            if i == 1 and r["Counter_Name"] == "TCC_REQ_sum":
                cnt[k] += 1; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
with open("gpurun_out/%s_l2_pmc.txt" % tag, "w") as o:
    for k in sorted(agg):
        n = max(cnt[k], 1)
        o.write("%s  launches %d  avg %.1f us\n" % (k, cnt[k], dur[k] / n))
        for c in sorted(agg[k]):
            o.write("    %-32s %16.0f per launch\n" % (c, agg[k][c] / n))
print(open("gpurun_out/%s_l2_pmc.txt" % tag).read())
PY
