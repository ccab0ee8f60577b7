#!/usr/bin/env python3
"""Compile the library's translation units (csrc/*.hip) with -Rpass-analysis=kernel-resource-usage and print one
line per kernel: VGPRs, AGPRs, SGPRs, scratch, LDS, occupancy (waves/SIMD).
Usage: tools/kernel_resources.py [regex]"""
import os, re, subprocess, sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "myldpccppapi_amd", "csrc")
from concurrent.futures import ThreadPoolExecutor


def remarks(src):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
           "-c", src, "-o", "/tmp/_kr_%s.o" % src, "-Rpass-analysis=kernel-resource-usage"]
    return subprocess.run(cmd, cwd=root, capture_output=True, text=True).stderr


# the library's translation units (csrc/Makefile: HIP_OBJS)
with ThreadPoolExecutor(7) as ex:
    out = "\n".join(ex.map(remarks, ["ldpc_hip.hip", "flood_sp.hip", "flood_ms.hip", "flood_ms16.hip", "engine_ldsp.hip", "engine_fused.hip", "engine_layered.hip"]))
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print("%-70s %5s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
for r in rows:
    nm = re.sub(r"^void ldpc::|\(.*$", "", r["name"])
    if pat and not pat.search(nm):
        continue
    print("%-70s %5s %5s %5s %7s %6s %4s" % (nm, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
                                            r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"),
                                            r.get("Occupancy [waves/SIMD]")))
