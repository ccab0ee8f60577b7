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
import csv, glob, collections, json
acc = collections.defaultdict(float)
for f in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "ldsp" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
stats = {}
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "ldsp" in row["Name"]:
            stats = {"kernel": row["Name"], "calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
bench = json.loads(open("$OUT/bench_trace.json").read().strip().splitlines()[-1])
frames, iters = bench["config"]["frames"], bench["config"]["avg_iterations_per_frame"]
out = {"what": "rocprofv3 on bench.py --config bg1_layered (layered_ldsp_kernel), counters are sums over ONE launch "
               "(8192 frames x 20 iterations), each set collected in a pass of its own",
       "kernel_trace": stats, "bench_line_under_trace": bench, "counters_per_launch": dict(acc)}
if "SQ_INSTS_VALU" in acc and stats:
    simd_cycles = acc["SQ_INSTS_VALU"] * 4
    out["derived"] = {
        "valu_wave_instructions_per_edge_iteration_x64": acc["SQ_INSTS_VALU"] / (frames * iters * 121344 / 64),
        "valu_busy_fraction_at_2.4GHz": simd_cycles / (1024 * stats["avg_ns"] * 2.4),
        "fabric_read_GB": acc.get("FETCH_SIZE", 0) * 2 * 1024 / 1e9, "fabric_write_GB": acc.get("WRITE_SIZE", 0) * 1024 / 1e9,
        "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); both counters include Infinity-Cache hits, "
                "i.e. they are L2<->fabric traffic (the check-record rings), not HBM traffic"}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out.get("derived"), indent=1))
PY
