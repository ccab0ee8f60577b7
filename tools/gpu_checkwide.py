#!/usr/bin/env python3
"""Plain (unlinked) check kernels: narrow waves in merged launches (default) against wide waves, on a
WiMAX code in the streaming sum-product path.  usage: gpu_checkwide.py [N] [frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2304
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K, M, z = codes.wimax_dims(0, N)
rows, cols = codes.wimax_edges(0, N)
g = L.Graph(rows, cols, M, N)
y = (1.0 + 0.95 * torch.randn(B, N, device="cuda", dtype=torch.float32))
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
for rep in range(2):
    for name, tune in (("narrow+merged", {}), ("wide", {"check_wide": True}), ("narrow unmerged", {"merge": False})):
        dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=40, frames_per_lane=4, tune=tune)
        for _ in range(2):
            dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, None)
        torch.cuda.synchronize()
        dec.set_timing(True)
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, None)
        torch.cuda.synchronize()
        st = dec.stats()
        print("%-16s total %.3f ms  check %.3f  var %.3f  other %.3f" % (name, st["ms_total"], st["ms_check"], st["ms_var"], st["ms_other"]), flush=True)
        dec.close()
