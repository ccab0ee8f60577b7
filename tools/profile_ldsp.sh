#!/bin/bash
# rocprofv3 passes over the BG1 layered configuration (layered_ldsp_kernel): kernel trace, then
# instruction-mix / HBM counters in passes of their own.  Outputs under gpurun_out/prof_ldsp/.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_ldsp
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config bg1_layered --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_WAVES" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$tag -- python3 $REPO/bench.py --config bg1_layered --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/pmc_$tag.err || echo "pass $tag failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        if "ldsp" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
    print(f.split("/")[-3] if "/" in f else f, dict(acc))
PY
