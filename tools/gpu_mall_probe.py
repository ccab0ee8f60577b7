#!/usr/bin/env python3
"""What the memory system sustains for a float4 copy (read + write counted) as a function of the
footprint: buffers that fit the 256 MiB Infinity Cache against buffers far beyond it.
usage: gpu_mall_probe.py  (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import myldpccppapi_amd as L

for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048):
    print("copy of %5d MiB (footprint %5d MiB): %7.1f GB/s" % (mb, 2 * mb, L.capi.hbm_probe(0, mb << 20, 20)), flush=True)
