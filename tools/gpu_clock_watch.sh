#!/bin/bash
# Sample the GPU's clocks, power and temperature (rocm-smi, read-only) while the headline step runs back to back
# for about 20 s from a cold start: does the 123 ms -> 137 ms change of regime coincide with a change of clocks?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/clock_watch.txt
: > $OUT
( for i in $(seq 1 140); do echo "t=$(date +%s.%N)" >> $OUT; rocm-smi --showclocks --showpower --showtemp --showperflevel 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature|Performance" >> $OUT; sleep 0.15; done ) &
W=$!
sleep 1
python3 - <<PY >> $REPO/gpurun_out/clock_watch_steps.txt
import sys, time, json
sys.path.insert(0, "$REPO")
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import channel, codes
N, K, B = 64800, 32400, 4096
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=50)
y = channel.awgn_device(N, 0, B, 0.95, seed=20260101)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for i in range(120):
    t0 = time.time()
    dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
    torch.cuda.synchronize()
    print("step %d t=%.3f ms=%.2f" % (i, t0, (time.time() - t0) * 1e3), flush=True)
PY
wait $W
