#!/bin/bash
# Kernel trace of one decode with hand-overs: which kernels ran when, on how many tiles (grid), and the gaps between them.
#   tools/trace_handover.sh <tag> <algo> <snr> [compact]
set -e
TAG=${1:-r03_handover}; ALGO=${2:-sp}; SNR=${3:-5.0}; COMPACT=${4:-auto}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/tools/gpu_handover_probe.py --algo $ALGO --snr=$SNR --compact=$COMPACT --poll 2 --batches 1 > $OUT/probe.log 2> $OUT/trace.err
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last decode call: everything after the last awgn kernel
last = max(i for i, r in enumerate(rows) if "awgn" in r["Kernel_Name"])
rows = rows[last + 1:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
with open("$OUT/timeline.txt", "w") as o:
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0][:70]
        o.write("%9.1f us  +gap %7.1f  dur %8.1f  grid %s/%s  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
                r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), name))
        prev_end = e
    o.write("total %.1f us, %d kernels\n" % ((prev_end - t0) / 1e3, len(rows)))
PY
rm -rf $OUT/trace
tail -1 $OUT/timeline.txt
