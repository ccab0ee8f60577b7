#!/usr/bin/env python3
"""Static instruction counts per kernel from the compiled ISA (hipcc -S, device only): vector ALU, the IEEE
division sequences among them (v_div_* / v_rcp_*), scalar, LDS, vector memory, scratch.
Usage: tools/kernel_isa_counts.py <translation unit, e.g. flood_sp.hip> [regex on the demangled name]"""
import os, re, subprocess, sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "myldpccppapi_amd", "csrc")
src = sys.argv[1] if len(sys.argv) > 1 else "flood_sp.hip"
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
out = "/tmp/_isa_%s.s" % os.path.basename(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
                       "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero",
                       "-S", "--cuda-device-only", "-o", out, src], cwd=root, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
print("%-72s %6s %6s %6s %5s %5s %7s" % ("kernel", "valu", "div", "salu", "lds", "vmem", "scratch"))
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if not m:
        continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void ", "", name).replace("ldpc::", "")
    name = re.sub(r"\((ldpc::)?\w+Args.*$", "", name)
    if pat and not pat.search(name):
        continue
    end = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
    ins = [x.strip() for x in lines[i:end] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
    cnt = lambda *p: sum(1 for x in ins if x.startswith(p))
    print("%-72s %6d %6d %6d %5d %5d %7d" % (name[:72], cnt("v_"), cnt("v_div", "v_rcp"), cnt("s_"), cnt("ds_"),
                                            cnt("global_", "buffer_", "flat_"), cnt("scratch_")))
