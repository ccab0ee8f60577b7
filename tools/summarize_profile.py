#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into
small committed files under profiles/:
  <tag>_kernel_stats.csv   per-kernel calls / total / average / min / max (from --kernel-trace --stats)
  <tag>_pmc_traffic.json   per-kernel FETCH_SIZE / WRITE_SIZE averages per launch (separate --pmc passes)
                           and the HBM bytes they imply, corrected as MI355X_MICROARCH.md (HBM section)
                           prescribes: counters are in KiB; FETCH_SIZE under-reports coalesced reads on
                           gfx950 (128-byte requests counted as 64 bytes).  The factor is CALIBRATED in the
                           same run, not assumed: bench.py's copy probe (hbm_probe_copy_kernel, both cache
                           policies) reads exactly `probe_bytes` per launch, so factor = known / counted; it
                           is applied to every kernel and recorded with the kernel's row, next to the
                           write side's own check (known / counted, expected 1).
  traffic.json             {kernel display name: corrected HBM bytes per launch} read by bench.py
usage: tools/summarize_profile.py <tag>"""
import csv, glob, json, os, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("ldpc::", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    if len(name) > 80:
        name = name[:77] + "..."
    return name


def display(name):
    """check_kernel<0, 7, 4> -> check_kernel<sp,7,4> (the names bench.py prints)"""
    m = re.match(r"(check_kernel|check_link_kernel|check_link_narrow_kernel|check_link_narrow2_kernel|var_kernel)<(\d), (\d+), (\d)(?:, \d)?, (float|_Float16)(?:, (\d))?>", name)
    if m:
        algo = ("sp", "ms")[int(m.group(2))] + ("16" if m.group(5) != "float" else "")
        kern = "check_link_half_kernel" if (m.group(1) == "check_link_narrow_kernel" and m.group(6) == "2") else m.group(1)
        return "%s<%s,%s,%s>" % (kern, algo, m.group(3), m.group(4))
    m = re.match(r"(check_group_kernel|var_group_kernel)<(\d), (\d), (float|_Float16), (\d+), (\d+)>", name)
    if m:
        algo = ("sp", "ms")[int(m.group(2))] + ("16" if m.group(4) != "float" else "")
        return "%s<%s,%s-%s,%s>" % (m.group(1), algo, m.group(5), m.group(6), m.group(3))
    m = re.match(r"layer_kernel<(\d+), (\d)>", name)
    if m:
        return "layer_kernel<layered,%s,%s>" % (m.group(1), m.group(2))
    return name


def newest(pattern):
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
rows = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        rows.append([short(r["Name"]), r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]),
                     r["Percentage"], r["MinNs"], r["MaxNs"]])
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras (MI355X, 1 GPU; the decoder's creation-time timings -- forms and placement candidates, about 50 launches of the check kernel and 45 of each variable-node kernel on zeroed arrays -- are in the counts)"])
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        w.writerows(rows)

pmc = {}
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = newest(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    acc = {}
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for k, (n, tot) in acc.items():
        pmc.setdefault(k, {})[counter + "_KiB_per_launch"] = tot / n
        pmc[k]["launches_" + counter] = n
traffic = {}
# calibration from the in-run copy probe: bench.py copies PROBE_BYTES per launch (read) and writes as many
PROBE_BYTES = float(1 << 30)
cal = {"fetch_factor": None, "write_factor": None, "probe_rows": {}}
ff, wf = [], []
for k, v in pmc.items():
    if k.startswith("hbm_probe_copy_kernel") and "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
        f = PROBE_BYTES / (v["FETCH_SIZE_KiB_per_launch"] * 1024)
        w = PROBE_BYTES / (v["WRITE_SIZE_KiB_per_launch"] * 1024)
        cal["probe_rows"][k] = {"known_bytes_read": PROBE_BYTES, "fetch_factor": f, "write_factor": w}
        ff.append(f)
        wf.append(w)
if ff:
    cal["fetch_factor"] = sum(ff) / len(ff)
    cal["write_factor"] = sum(wf) / len(wf)
fetch_factor = cal["fetch_factor"] if cal["fetch_factor"] else 2.0          # the guide's figure when no probe ran
for k, v in pmc.items():
    if "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
        fetch = v["FETCH_SIZE_KiB_per_launch"] * 1024 * fetch_factor
        write = v["WRITE_SIZE_KiB_per_launch"] * 1024
        v["fetch_correction"] = round(fetch_factor, 4)
        v["fetch_correction_source"] = "copy probe of the same run" if cal["fetch_factor"] else "MI355X_MICROARCH.md (no probe in this run)"
        v["hbm_bytes_per_launch"] = fetch + write
        traffic[display(k)] = int(fetch + write)
if pmc:
    # which launch shape the byte counts belong to (bench.py scales them per frame and says so)
    cfg = {"source": "profiles/%s_pmc_traffic.json" % tag}
    try:
        line = [l for l in open(os.path.join(src, "bench_pmc_fetch.json")) if l.startswith("{")][-1]
        cfg["frames_per_gpu"] = json.loads(line)["config"]["frames_per_gpu"]
    except Exception:
        cfg["frames_per_gpu"] = 4096
    cfg["fetch_calibration"] = cal
    traffic["__config__"] = cfg
    try:        # keep the sections other tools wrote (VALU counts of the record kernel: tools/profile_ldsp.sh)
        prev = json.load(open(os.path.join(dst, "traffic.json")))
        for key in prev:
            if key.startswith("__") and key != "__config__":
                traffic[key] = prev[key]
    except Exception:
        pass
    pmc["__calibration__"] = cal
    json.dump(pmc, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1, sort_keys=True)
for k in sorted(pmc):
    print(k, pmc[k])
print('calibration', cal)
