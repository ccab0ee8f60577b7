#!/bin/bash
# Kernel timeline of ONE step of configs[4] (rate 9/10, fp16 messages, early termination with polling and hand-overs).
set -e
TAG=${1:-r03_config4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --config dvbs2_910_f16 --steps 1 --warmup 2 > $OUT/bench.json 2> $OUT/trace.err
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last init_kernel on
last = max(i for i, r in enumerate(rows) if "init_kernel" in r["Kernel_Name"])
rows = rows[last:]
end = max(i for i, r in enumerate(rows) if "summary_kernel" in r["Kernel_Name"])
rows = rows[:end + 1]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
busy = {}
with open("$OUT/timeline.txt", "w") as o:
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void ldpc::", "").split("(")[0][:60]
        o.write("%9.1f us  +gap %6.1f  dur %7.1f  grid %s/%s  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), name))
        key = name.split("<")[0]
        b = busy.setdefault(key, [0, 0.0, 0.0]); b[0] += 1; b[1] += (e - s) / 1e3; b[2] += max(0, s - prev_end) / 1e3
        prev_end = e
    o.write("total %.1f us, %d kernels\n" % ((prev_end - t0) / 1e3, len(rows)))
    for k, b in sorted(busy.items(), key=lambda kv: -kv[1][1]):
        o.write("%-40s n=%3d  busy %8.1f us  gap before %7.1f us\n" % (k, b[0], b[1], b[2]))
PY
rm -rf $OUT/trace
tail -25 $OUT/timeline.txt
