#!/bin/bash
# L2<->fabric traffic and vector-memory read instructions of layered_ldsp_kernel for several builds of the library:
# tools/pmc_ldsp_variants.sh lib1.so lib2.so ...   ("" = the library in the tree)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_ldsp_variants
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for lib in "$@"; do
  i=$((i+1))
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    tag=v${i}_$(echo $set | tr ' ' '_' | cut -c1-30)
    LDPC_HIP_LIB=${lib:+$REPO/$lib} rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $REPO/bench.py --config bg1_layered --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/$tag.err || echo "pass $tag failed"
  done
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float)
for f in sorted(glob.glob("$OUT/v${i}_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "ldsp" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
print("lib=${lib:-current}", {k: "%.4g" % v for k, v in acc.items()}, "fabric read GB %.1f write GB %.1f" % (acc.get("FETCH_SIZE", 0) * 2048 / 1e9, acc.get("WRITE_SIZE", 0) * 1024 / 1e9))
PY
done
