#!/bin/bash
# One look at the column-fused check kernel with memory-system counters, on whichever state the box is in (the step has a
# 123 ms and a 137 ms regime that differ almost only in this kernel).  Three rocprofv3 --pmc passes of
# one bench step each (own processes; the kernel trace of each pass tells which regime that pass was in).
set -e
TAG=${1:-r03_regime}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_WAVES" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL" \
           "TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL" \
           "TCC_HIT TCC_MISS TCC_TOO_MANY_EA_WRREQS_STALL TCC_BUSY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench$i.json 2> $OUT/pass$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for i in range(1, 5):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("$OUT/pass%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "check_link" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    dur = []
    for f in glob.glob("$OUT/pass%d/**/*kernel_trace.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            if "check_link_kernel" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    n = max(1, len(dur))
    out["pass%d" % i] = {"check_link_avg_ms": sum(dur) / n, "launches": len(dur),
                         "per_launch": {k: v[1] / max(1, len(dur)) for k, v in acc.items()}}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
