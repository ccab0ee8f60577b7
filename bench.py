#!/usr/bin/env python3
"""Headline benchmark: decoded info Mbit/s of batched LDPC BP decoding on MI355X.

Workload (BASELINE.json configs[1], per GPU): DVB-S2-profile (64800, 32400) rate 1/2,
batch 4096 frames, 50 iterations, sum-product fp32 (probability domain, the reference's
arithmetic), inputs resident in HBM.  One "step" = one full decode of the batch through
the C ABI (`ldpc_decode_device`) on torch's current stream.  The channel is all-zero
codeword + AWGN at sigma = 0.95: no frame's syndrome becomes clean, so all 50 rounds do
full work with the reference's freeze-on-clean-syndrome semantics switched ON.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one rank per GPU (torch.distributed, backend "nccl" = RCCL).  Invoked plainly
(`python bench.py --gpus N`, no WORLD_SIZE in the environment) this script starts the N
ranks itself -- as fresh child processes of `python -m torch.distributed.run`, before
anything in this process touches the GPU -- relays rank 0's JSON line and exits with the
children's status; invoked under torch.distributed.run it is one of the ranks.  Frames
shard across ranks with no exchange while decoding; the decoded bytes are all-gathered at
the end of every step (weak scaling: 4096 frames per GPU).

Prints ONE JSON line (rank 0): metric/value/unit (whole-job Mbit/s), ms_per_step,
`roofline` for the dominant kernel (HIP-event time on the launch stream, live) and, at
N = 1, `cpu_baseline` (the oracle's restatement of the reference's CPU decoder,
MyLdpc.cpp:684-784, on a bounded sample), `extra` (BASELINE.json configs[3] and [4] at
their full batch sizes), `ber` (BER @ SNR points, the other half of BASELINE's metric), `host_path` (the reference's own
signature: host buffers in, host buffers out, PCIe included, in both input modes), `host_path_devices`
(the same call on ONE handle over every visible GPU: `ldpc_decoder_create_multi`) and `coder_path` (the
C++ class itself: Test.cpp's call sequence through libmyldpc.so, timed by `CoderBench`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CODE, K_CODE = 64800, 32400
BATCH_PER_GPU = 4096
ITERS = 50
SIGMA = 0.95
SEED = 20260101
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4     # G wave64 VALU instructions per second: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles

# Extra measurement points (BASELINE.json configs[3], configs[4]); reported under "extra" of the
# headline line at N = 1, or alone with   python bench.py --config bg1_layered | dvbs2_910_f16
EXTRA_CONFIGS = {
    "bg1_layered": dict(desc="5G-NR BG1-profile QC code, Z=384 (N=26112, K=8448, E=121344), batch 8192, "
                             "layered min-sum fp32, 20 iterations, sigma=1.1 (no frame converges)",
                        batch=8192, iters=20, sigma=1.1, algo="layered", msg="f32", early=True, poll=0),
    "dvbs2_910_f16": dict(desc="DVB-S2-profile (64800,58320) rate-9/10 (E=194399, check degree 30), batch 4096, "
                               "flooding min-sum with fp16 messages, early termination (syndrome) on, "
                               "max 50 iterations, sigma=0.43",
                          batch=4096, iters=50, sigma=0.43, algo="ms", msg="f16", early=True, poll=2),
}


def measure_extra(name, steps, warmup, batch=0, sigma=0.0, fpl=0, poll=-1, tune=None):
    """One extra config on the current GPU: dict with the headline fields of its own."""
    import numpy as np
    import torch
    import myldpccppapi_amd as L
    from myldpccppapi_amd import channel, codes
    c = EXTRA_CONFIGS[name]
    if name == "bg1_layered":
        Z = 384
        rows, cols = codes.nr_bg1_profile_edges(Z)
        N, K, M, layer = 68 * Z, 22 * Z, 46 * Z, Z
        bytes_fi = 16 * len(rows)
    else:
        N, K = 64800, 58320
        rows, cols = codes.dvbs2_profile_edges(N, K)
        M, layer = N - K, 0
        bytes_fi = 8 * len(rows) + 2 * N
    B = batch or c["batch"]
    if sigma:
        c = dict(c, sigma=sigma, desc=c["desc"] + " [sigma override %.3f]" % sigma)
    if poll >= 0:
        c = dict(c, poll=poll, desc=c["desc"] + " [poll_interval %d]" % poll)
    g = L.Graph(rows, cols, M, N)
    dec = L.Decoder(g, K, max_batch=B, algo=c["algo"], max_iter=c["iters"], early_term=c["early"],
                    layer_rows=layer, msg_dtype=c["msg"], poll_interval=c["poll"], frames_per_lane=fpl, tune=tune)
    y = channel.awgn_device(N, 0, B, c["sigma"], seed=SEED)
    out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
    it = torch.empty(B, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), s)
    torch.cuda.synchronize()
    dec.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = dec.stats()
    kt = [k for k in dec.kernel_times() if k["phase"] in (0, 1, 2)]
    iters = it.cpu().numpy()
    frame_iters = float(iters.sum())
    kms = sum(k["ms_total"] for k in kt)
    one_launch = len(kt) == 1 and ("ldsp" in kt[0]["name"] or "fused" in kt[0]["name"])
    res = {"metric": "decoded Mbit/s (info bits)", "value": round(B * K / dt / 1e6, 2), "unit": "Mbit/s",
           "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 3),
           "higher_is_better": True, "dtype": "f16" if c["msg"] == "f16" else "f32", "data": "synthetic",
           "config": {"workload": c["desc"], "frames": B, "rounds_launched": st["iterations_launched"],
                      "avg_iterations_per_frame": round(frame_iters / B, 2),
                      "frames_converged": st["frames_converged"],
                      "bit_errors_in_converged_frames": int(np.unpackbits(out.cpu().numpy().reshape(B, -1)[iters < c["iters"]]).sum())}}
    res["config"]["placement"] = dec.placement()      # streaming decoders with large arrays: what the creation-time search saw
    if one_launch:
        # LDS / cache resident decode (one launch; frame state in LDS, check records in L2 / Infinity Cache): HBM sees
        # the channel values in and the packed bits out only, so HBM is not what bounds it.  The kernel is bound by
        # vector-ALU issue: achieved = VALU wave-instructions per second, peak = CUs x 4 SIMDs x clock / 4 cycles per
        # wave64 instruction.  The instruction count per frame-iteration comes from the committed PMC profile of this
        # kernel (SQ_INSTS_VALU over one launch / frames / iterations: the code path per frame-iteration is fixed).
        prof = {}
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("__valu__", {}).get(name, {})
        except Exception:
            pass
        peak = VALU_PEAK_GINST
        per_fi = prof.get("valu_wave_insts_per_frame_iteration")
        ach = None if not per_fi else per_fi * frame_iters * steps / (kms * 1e-3) / 1e9
        res["roofline"] = {"bound": "valu", "kernel": kt[0]["name"], "unit": "G wave-instructions/s", "peak": peak,
                           "peak_is": "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction",
                           "achieved": None if ach is None else round(ach, 1),
                           "frac": None if ach is None else round(ach / peak, 4),
                           "valu_wave_insts_per_frame_iteration": per_fi, "valu_source": prof.get("source"),
                           "avg_launch_ms": round(kms / max(1, kt[0]["launches"]), 4),
                           "edge_updates_per_s_G": round(len(rows) * frame_iters * steps / (kms * 1e-3) / 1e9, 1),
                           "hbm_side_note": {"bytes_per_launch": int(kt[0]["bytes_total"] / max(1, kt[0]["launches"])),
                                             "GB/s": round(kt[0]["bytes_total"] / (kms * 1e-3) / 1e9, 1),
                                             "what": "channel values in + packed bits out: all the HBM traffic of the launch; an "
                                                     "HBM-streaming formulation of the same schedule would move %d B per "
                                                     "frame-iteration (16 E)" % bytes_fi}}
    else:
        # streaming kernels under early termination: a tile that was finished when a round began leaves at kernel
        # entry, so a round's traffic is priced at the frames of the tiles that still worked (counted on the device:
        # stats.frame_rounds), not at frames x rounds launched
        fr = st["frame_rounds"] * steps
        kb = bytes_fi * fr
        res["roofline"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "bytes_per_frame_iteration": bytes_fi, "frame_rounds_worked_per_step": st["frame_rounds"],
                           "frame_rounds_if_no_tile_skipped": st["iterations_launched"] * B,
                           "frame_iterations_needed": int(frame_iters),
                           "achieved": round(kb / (kms * 1e-3) / 1e9, 1),
                           "frac": round(kb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "whole_step_frac": round(bytes_fi * st["frame_rounds"] / dt / 1e9 / HBM_PEAK_GBS, 4),
                           "per_kernel_avg_ms": {k["name"]: round(k["ms_total"] / k["launches"], 4)
                                                 for k in sorted(kt, key=lambda k: -k["ms_total"])[:6]}}
    dec.close()
    del y, out, it
    torch.cuda.empty_cache()
    return res


def host_path(g, groups=3, B=BATCH_PER_GPU, devices=None, modes=("staged", "lock_pages"), gen_device=0):
    """The reference's own signature (`Coder::decode` -> `ldpc_decode`): host buffers in, host
    buffers out, `groups` x B frames per device, sum-product at full work -- PCIe transfers included.
    devices = None: one decoder on the current GPU; a list: ONE handle over those GPUs
    (`ldpc_decoder_create_multi`), the frame stream cut into one contiguous range per device."""
    import numpy as np
    import torch
    import myldpccppapi_amd as L
    from myldpccppapi_amd import channel
    ndev = len(devices) if devices else 1
    frames = groups * B * ndev
    y_host = torch.empty((frames, N_CODE), dtype=torch.float32)         # pageable, as a caller's malloc
    for k in range(groups * ndev):
        y_host[k * B:(k + 1) * B] = channel.awgn_device(N_CODE, k * B, B, SIGMA, seed=SEED, device=gen_device).cpu()
    torch.cuda.empty_cache()
    ynp = y_host.numpy()
    res = {"frames": frames, "max_batch": B, "devices": list(devices) if devices else None, "unit": "Mbit/s"}
    first = None
    for mode in modes:
        dec = L.Decoder(g, K_CODE, max_batch=B, algo="sp", max_iter=ITERS, llr_scale=8.0, early_term=True,
                        device=gen_device, devices=devices, host_input=mode)
        dec.decode(ynp[:64 * ndev], want_iters=False)                     # staging slots, first-touch
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            out, _ = dec.decode(ynp, want_iters=False)
            times.append(time.perf_counter() - t0)
        dec.close()                                                       # raises if a page-locked block was left behind
        if first is None:
            first = out
        dt = min(times)
        res[mode] = {"value": round(frames * K_CODE / dt / 1e6, 2), "ms": round(dt * 1e3, 2),
                     "first_call_ms": round(times[0] * 1e3, 2), "calls": len(times),
                     "same_bytes_as_first_mode": bool(np.array_equal(out, first))}
    res["value"] = res[modes[0]]["value"]
    res["locked_ranges_left"] = list(L.capi.host_locked_ranges())
    res["what"] = ("ldpc_decode (Coder::decode's signature): %d frames from pageable host memory, %d groups of %d per "
                   "device, sum-product fp32, %d iterations at full work, packed bytes back in host memory; the copy-in of "
                   "group k+1 and the copy-out of group k-1 overlap the decode of group k; best of 3 calls on the same buffer.  "
                   "staged (the default) = the handle's copy threads move each group through the library's pinned ring; "
                   "lock_pages (opt-in) = the caller's pages are page-locked for the call and read in place"
                   % (frames, groups, B, ITERS))
    return res


def coder_path(frames=BATCH_PER_GPU, iters=40, snr=2.6, repeat=3, timeout=600):
    """The C++ class as a Test.cpp user sees it (Test.cpp:47-64,105-112): Coder(32400, 64800, rate_1_2)
    -- the reference-constructible code of the headline size, z = 2700, E = 205200 -- forEncoder, encode,
    test, addDecodeType(DecodeSP), decode of `frames` frames at the reference's 40 iterations, through
    libmyldpc.so in a process of its own (myldpccppapi_amd/CoderBench), wall clock."""
    exe = os.path.join(ROOT, "myldpccppapi_amd", "CoderBench")
    if not os.path.exists(exe):
        return {"error": "%s not built" % exe}
    cmd = [exe, "0", str(N_CODE), str(frames), str(frames), str(snr), "SP", "--iters", str(iters), "--repeat", str(repeat)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    if p.returncode != 0:
        return {"error": "rc %d: %s" % (p.returncode, (p.stderr or p.stdout)[-400:])}
    kv = {}
    for ln in p.stdout.splitlines():
        if "=" in ln:
            k, v = ln.split("=", 1)
            kv[k.strip()] = v.strip()
    dec = sorted(float(v) for k, v in kv.items() if k.startswith("decode_s["))
    first = float(kv.get("decode_s[0]", "nan"))
    return {"value": float(kv["decode_info_mbit_s"]), "unit": "Mbit/s", "frames": frames, "batchSize": frames,
            "iterations": iters, "decodeType": "DecodeSP", "K": int(kv["K"]), "N": int(kv["N"]), "z": int(kv["z"]),
            "NonZeros": int(kv["NonZeros"]), "sd": float(kv["sd"]), "Time": int(kv["Time"]), "ErrNum": int(kv["ErrNum"]),
            "ThroughPut_bytes_per_s": float(kv["ThroughPut"]), "decode_ms_best": round(dec[0] * 1e3, 2),
            "decode_ms_first_call": round(first * 1e3, 2), "decode_calls": len(dec),
            "forEncoder_ms": round(float(kv["forEncoder_s"]) * 1e3, 3), "encode_ms": round(float(kv["encode_s"]) * 1e3, 2),
            "encode_info_mbit_s": float(kv["encode_info_mbit_s"]), "test_ms": round(float(kv["test_s"]) * 1e3, 1),
            "addDecodeType_ms": round(float(kv["addDecodeType_s"]) * 1e3, 1),
            "what": "CoderBench 0 %d %d %d %.1f SP --iters %d: Test.cpp's sequence on Coder(32400, 64800, rate_1_2); decode() = "
                    "ldpc_decode with poll_interval 4 from malloc'ed buffers (default input mode: staged), PCIe included; "
                    "ThroughPut = info bytes per wall-clock second of the best of %d decode() calls (the reference prints "
                    "CPU seconds, Test.cpp:111); at 2.6 dB the reference's fixed exp(8 y) sum-product leaves frames unconverged (ErrNum > 0, "
                    "Time = 40): the call does full work, as the headline step does; encode() is the structured O(E) encoder on "
                    "z-bit words, the frames spread over the host's threads"
                    % (N_CODE, frames, frames, snr, iters, repeat)}


def ber_points(g, B=BATCH_PER_GPU):
    """BER @ SNR on the headline code, reference convention (Test.cpp:56-57: BPSK +-1, sd =
    10^(-SNR_dB/20)), all-zero codeword, channel and error count on the GPU: one batch per point."""
    import torch
    import myldpccppapi_amd as L
    from myldpccppapi_amd import channel
    pts = []
    y = torch.empty((B, N_CODE), dtype=torch.float32, device="cuda")
    out = torch.empty(L.out_bytes(K_CODE, B), dtype=torch.uint8, device="cuda")
    it = torch.empty(B, dtype=torch.int32, device="cuda")
    for algo, snrs in (("sp", (3.0, 4.0)), ("ms", (1.5, 1.7))):
        dec = L.Decoder(g, K_CODE, max_batch=B, algo=algo, max_iter=ITERS, llr_scale=8.0, early_term=True, poll_interval=2)
        for snr in snrs:
            sd = 10.0 ** (-snr / 20.0)
            channel.awgn_device(N_CODE, 0, B, sd, seed=SEED + 1, out=y)
            t0 = time.perf_counter()
            dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), None)
            bit_err, byte_err, frame_err = channel.count_errors_device(out, None, B)
            dt = time.perf_counter() - t0
            pts.append({"algo": algo, "snr_db": snr, "sd": round(sd, 4), "frames": B, "info_bits": B * K_CODE,
                        "bit_errors": bit_err, "ber": bit_err / (B * K_CODE), "byte_errors": byte_err,
                        "fer": frame_err / B, "avg_iterations": round(float(it.float().mean()), 2),
                        "decode_mbit_s": round(B * K_CODE / dt / 1e6, 1)})
        dec.close()
    del y, out, it
    torch.cuda.empty_cache()
    return {"code": "DVB-S2-PROFILE surrogate (64800,32400) -- the standard's structure from a seeded table, not "
                    "its Annex B addresses: error rates are this table's, not DVB-S2's",
            "max_iter": ITERS, "note": "sp = the reference's probability-domain decoder with its fixed exp(8 y) scale "
            "(decodeCL.c:9); ms = flooding min-sum (decodeCPU's arithmetic)", "points": pts}


def cpu_baseline(rows, cols, seconds_budget=20.0, gpu_graph=None, gpu_y=None):
    """Reference CPU decode (min-sum, MyLdpc.cpp:684-784) via the oracle port, all host
    cores (frames split over threads; the C call releases the GIL) and one core."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    import oracle
    M = N_CODE - K_CODE
    g = oracle.Graph(rows, cols, M, N_CODE, K_CODE)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:        # a container's CPU share (cgroup v2 quota) can be smaller than its affinity mask
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    cores = max(1, min(cores, 64))
    # one frame first: calibrates the sample size (~0.3 s per frame-50-iterations per core)
    y1 = oracle.awgn(N_CODE, 0, 1, SIGMA, seed=SEED)
    t0 = time.perf_counter()
    oracle.decode(g, y1, "ms", max_iter=ITERS)
    t_one = time.perf_counter() - t0
    per_thread = max(1, min(8, int(seconds_budget / max(t_one, 1e-3) / cores)))
    frames = per_thread * cores
    y = oracle.awgn(N_CODE, 0, frames, SIGMA, seed=SEED)      # = the first frames of the GPU's batch
    chunks = [y[i * per_thread:(i + 1) * per_thread] for i in range(cores)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        outs = list(ex.map(lambda c: oracle.decode(g, c, "ms", max_iter=ITERS), chunks))
    dt = time.perf_counter() - t0
    # the same frames, the same algorithm on the GPU: bytes and iteration counts must be identical
    same = None
    if gpu_graph is not None:
        import myldpccppapi_amd as L
        dec = L.Decoder(gpu_graph, K_CODE, max_batch=frames, algo="ms", max_iter=ITERS)
        got, git = dec.decode(gpu_y[:frames].cpu().numpy())
        same = bool(np.array_equal(got, np.concatenate([o["out"] for o in outs])) and
                    np.array_equal(git, np.concatenate([o["iters"] for o in outs])) and
                    np.array_equal(gpu_y[:frames].cpu().numpy().view(np.uint32), y.view(np.uint32)))
        dec.close()
    return {
        "value": round(frames * K_CODE / dt / 1e6, 4), "unit": "Mbit/s", "cores": cores, "kind": "port",
        "one_core_mbit_s": round(K_CODE / t_one / 1e6, 4),
        "same_frames_on_gpu_identical": same,
        "sample": "%d frames of the same code and noise (sigma=%.2f), %d iterations of the reference's CPU "
                  "min-sum decoder (oracle port of MyLdpc.cpp:684-784), %d threads x %d frames, %.1f s wall; "
                  "they are the first frames of the GPU's batch (same floats), and the GPU's min-sum decode of "
                  "them is compared byte for byte" % (frames, SIGMA, ITERS, cores, per_thread, dt),
    }


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh children.  Nothing
    in this process has touched the GPU (torch is not even imported yet)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it here
    env["LDPC_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    for ln in p.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if p.returncode != 0 or not lines:
        sys.exit(p.returncode or 1)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU (default: the config's)")
    ap.add_argument("--config", default="dvbs2_sp", choices=["dvbs2_sp"] + sorted(EXTRA_CONFIGS),
                    help="dvbs2_sp = the headline workload (default); others are extra measurement points")
    ap.add_argument("--algo", default="sp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip `extra` and `host_path` (profiling runs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 flow on fewer GPUs than ranks (collectives on CPU copies)")
    ap.add_argument("--sigma", type=float, default=0.0, help="extra configs: noise level override")
    ap.add_argument("--fpl", type=int, default=0, help="frames per lane override (tuning)")
    ap.add_argument("--poll", type=int, default=-1, help="extra configs: poll_interval override (0 = asynchronous)")
    ap.add_argument("--tune", default="", help='tuning fields as JSON, e.g. \'{"merge": false}\' (A/B experiments)')
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    # a CPU-side group for waiting WITHOUT a spinning kernel on the GPU (rank 0's device-list measurement at the end)
    cpu_group = dist.new_group(backend="gloo") if world > 1 else None

    if args.config != "dvbs2_sp":
        if world > 1:
            sys.exit("extra configs are single-GPU measurements")
        print(json.dumps(measure_extra(args.config, args.steps, args.warmup, args.batch, args.sigma, args.fpl, args.poll,
                                       json.loads(args.tune) if args.tune else None)), flush=True)
        return

    import myldpccppapi_amd as L
    from myldpccppapi_amd import channel, codes, sharding

    B = args.batch or BATCH_PER_GPU
    rows, cols = codes.dvbs2_profile_edges(N_CODE, K_CODE)
    g = L.Graph(rows, cols, N_CODE - K_CODE, N_CODE)
    dec = L.Decoder(g, K_CODE, max_batch=B, algo=args.algo, max_iter=ITERS, llr_scale=8.0,
                    early_term=True, device=local_rank, frames_per_lane=args.fpl,
                    tune=json.loads(args.tune) if args.tune else None)
    # synthetic channel: all-zero codeword + AWGN, generated in HBM, distinct per rank
    lo, hi = sharding.shard_range(B * world, rank, world)
    # (counter-based noise, csrc/ldpc_channel.h: frame lo + i of the seed's stream, whatever the world size)
    y = channel.awgn_device(N_CODE, lo, B, SIGMA, seed=SEED, device=local_rank)
    out = torch.empty(L.out_bytes(K_CODE, B), dtype=torch.uint8, device="cuda")
    gathered = torch.empty(world * out.numel(), dtype=torch.uint8, device="cuda") if world > 1 else None

    def step():
        s = torch.cuda.current_stream().cuda_stream
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
        if world > 1 and not rehearsal:
            dist.all_gather_into_tensor(gathered, out)     # the only collective: decoded bytes
        elif world > 1:
            parts = [torch.empty(out.numel(), dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, out.cpu())
            gathered.copy_(torch.cat(parts))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # per-launch HIP events on the launch stream, inside the timed region: on two of the K steps (the
    # events cost 2 % of a step, tools/gpu_timing_overhead.py), i.e. >= 100 launches of every kernel
    dec.set_timing(max(1, args.steps // 2))
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt_own = time.perf_counter() - t0
    dt = dt_own
    per_rank_ms = [dt_own / args.steps * 1e3]
    if world > 1:
        dev = "cpu" if rehearsal else "cuda"
        t = torch.tensor([dt_own], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        alls = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(alls, torch.tensor([dt_own], device=dev, dtype=torch.float64))
        per_rank_ms = [float(a.item()) / args.steps * 1e3 for a in alls]
        # the gathered bytes really are every rank's output: this rank's slice equals its own buffer
        assert torch.equal(gathered[rank * out.numel():(rank + 1) * out.numel()], out)

    st = dec.stats()
    kt = dec.kernel_times()
    link_form = dec.link_form()          # which form of the column-fused check kernel the creation-time timing picked
    placement = dec.placement()          # ... and on which of the candidate sets of arrays it runs
    assert st["iterations_launched"] == ITERS
    launch_frames = B
    flood = [k for k in kt if k["phase"] in (0, 1)]
    dom = max(flood, key=lambda k: k["ms_total"])
    # what this rank's box delivers right now: float4 copy of 1 GiB, the better of the default cache policy and
    # the streaming kernels' non-temporal one -- on EVERY rank, at the same moment, so that a GPU in a slow
    # state shows in the record (ranks.per_rank)
    try:
        probe, probe_default, probe_nt = L.capi.hbm_probe(local_rank, 1 << 30, 5, by_policy=True)
    except Exception:
        probe = probe_default = probe_nt = None
    mine = {"rank": rank, "device": local_rank, "ms_per_step": round(dt_own / args.steps * 1e3, 3),
            "kernel": dom["name"], "kernel_form": None if not link_form else link_form["form"], "placement": placement, "avg_launch_ms": round(dom["ms_total"] / dom["launches"], 4),
            "hbm_probe_gbs": None if probe is None else round(probe, 1)}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank == 0:
        frames_total = B * world * args.steps
        value = frames_total * K_CODE / dt / 1e6
        dom_avg_ms = dom["ms_total"] / dom["launches"]
        # bytes the dominant kernel's own loads and stores move per launch (for the column-fused check
        # kernel: less than its share of 16 E + 4 N, the fused columns' messages stay in registers) ...
        dom_bytes = dom["bytes_moved"] / dom["launches"]
        achieved = dom_bytes / (dom_avg_ms * 1e-3) / 1e9
        # ... and its share of SURVEY 8(d)'s two-kernel figure, for comparison
        dom_alg = dom["bytes_total"] / dom["launches"]
        all_bytes = sum(k["bytes_total"] for k in flood)
        all_moved = sum(k["bytes_moved"] for k in flood)
        all_ms = sum(k["ms_total"] for k in flood)
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                tc = tj.get("__config__", {})
                per = tj.get(dom["name"])
                if per is not None and tc.get("frames_per_gpu"):
                    traffic = int(per * launch_frames / tc["frames_per_gpu"])         # linear in the frames of a launch
                    fcal = (tc.get("fetch_calibration") or {}).get("fetch_factor")
                    traffic_note = ("PMC FETCH_SIZE x %s + WRITE_SIZE per launch from %s (rocprofv3, separate passes, "
                                    "%d frames per launch%s)" % ("%.3f (calibrated on the run's copy probe)" % fcal if fcal else "2",
                                                                 tc.get("source", "profiles/"), tc["frames_per_gpu"],
                                                                 "" if tc["frames_per_gpu"] == launch_frames else ", scaled to %d" % launch_frames))
            except Exception:
                traffic = None
        # what the box SUSTAINS right now, the GPU still hot from the timed steps: the same copy back to
        # back for 0.3 s (these boxes drop to about 5.2 TB/s under load at times; a burst does not see it)
        try:
            sustained = L.capi.hbm_sustained(local_rank, 1 << 30, 300)
        except Exception:
            sustained = None
        step_alg = (16 * g.E + 4 * N_CODE) * ITERS * B
        res = {
            "metric": "decoded Mbit/s (info bits), DVB-S2 N=64800 rate-1/2, 50 iters",
            "value": round(value, 2), "unit": "Mbit/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "DVB-S2-profile (64800,32400) rate-1/2 (seeded IRA table with the standard's "
                            "structure, E=226799), batch %d frames per GPU, %d iterations, sum-product fp32 "
                            "(probability domain), all-zero codeword + AWGN sigma=%.2f (no frame converges), "
                            "early termination on, inputs resident in HBM" % (B, ITERS, SIGMA),
                "global_batch": B * world, "frames_per_gpu": B, "iterations": ITERS, "algo": args.algo,
                "frames_per_launch": launch_frames,
                "parallelism": "frames sharded over %d GPU(s), all-gather of decoded bytes" % world,
                "coded_mbit_s": round(value * N_CODE / K_CODE, 2),
                "frames_converged": st["frames_converged"],
            },
            "ranks": {"world_size": dist.get_world_size() if world > 1 else 1,
                      "backend": ("gloo (rehearsal)" if rehearsal else "nccl (RCCL)") if world > 1 else None,
                      "self_launched": os.environ.get("LDPC_BENCH_SELF_LAUNCHED") == "1",
                      "ms_per_step_min": round(min(per_rank_ms), 3), "ms_per_step_max": round(max(per_rank_ms), 3),
                      # one entry per rank: its own step time, its dominant kernel's HIP-event launch time and the
                      # copy rate its GPU delivered right after the timed steps
                      "per_rank": per_rank},
            "roofline": {
                "bound": "hbm", "kernel": dom["name"], "kernel_form": link_form, "placement": placement, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_note,
                "avg_launch_ms": round(dom_avg_ms, 4), "launches": dom["launches"],
                "bytes_per_launch": int(dom_bytes),
                "bytes_are": "what the kernel's own loads and stores move (every Q of its rows and the fused "
                             "columns' channel values in, R of the unfused edges and the fused columns' new Q out)",
                # the same launch priced at its share of SURVEY 8(d)'s 16 E + 4 N (round 1's `frac`; ADVICE r1 asked
                # for the moved bytes in `frac` and this figure under its own key)
                "algorithmic_achieved": round(dom_alg / (dom_avg_ms * 1e-3) / 1e9, 1),
                "algorithmic_frac": round(dom_alg / (dom_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "two_kernel_formulation": {      # the same launch priced at its share of 16 E + 4 N
                    "bytes_per_launch": int(dom_alg), "achieved": round(dom_alg / (dom_avg_ms * 1e-3) / 1e9, 1),
                    "frac": round(dom_alg / (dom_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                # float4 copy of 1 GiB in this process: the better of the default cache policy and the
                # streaming kernels' non-temporal one (both listed)
                "hbm_probe_gbs": None if probe is None else round(probe, 1),
                "hbm_probe_by_policy": None if probe is None else {"default": round(probe_default, 1),
                                                                   "nontemporal": round(probe_nt, 1)},
                "frac_of_probe": None if not probe else round(achieved / probe, 4),
                "hbm_sustained_gbs": None if not sustained else round(sustained, 1),
                "frac_of_sustained": None if not sustained else round(achieved / sustained, 4),
                "all_flooding_kernels": {
                    "moved_achieved": round(all_moved / (all_ms * 1e-3) / 1e9, 1),
                    "moved_frac": round(all_moved / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "algorithmic_achieved": round(all_bytes / (all_ms * 1e-3) / 1e9, 1),
                    "algorithmic_frac": round(all_bytes / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "bytes_per_frame_iteration": 16 * g.E + 4 * N_CODE,
                    "moved_bytes_per_frame_iteration": int(all_moved / (dom["launches"] * launch_frames)),
                    "per_kernel": {k["name"]: {"avg_ms": round(k["ms_total"] / k["launches"], 4),
                                               "GB/s": round(k["bytes_moved"] / (k["ms_total"] * 1e-3) / 1e9, 1)}
                                   for k in flood},
                },
                # (16 E + 4 N) x iterations x frames over the whole step's wall time
                "whole_step_frac": round(step_alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                "whole_step_frac_of_probe": None if not probe else round(step_alg / (dt / args.steps) / 1e9 / probe, 4),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(rows, cols, gpu_graph=g, gpu_y=y)
        if world == 1 and not args.no_extras:
            dec.close()
            del y, out
            torch.cuda.empty_cache()
            extra = {}
            for name, key in (("bg1_layered", "bg1_layered@8192"), ("dvbs2_910_f16", "dvbs2_910_f16@4096")):
                try:
                    r = measure_extra(name, steps=max(3, args.steps), warmup=2)
                    extra[key] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                                  "dtype": r["dtype"], "workload": r["config"]["workload"],
                                  "avg_iterations_per_frame": r["config"]["avg_iterations_per_frame"],
                                  "frames_converged": r["config"]["frames_converged"],
                                  "bit_errors_in_converged_frames": r["config"]["bit_errors_in_converged_frames"],
                                  "roofline": r["roofline"]}
                except Exception as e:      # an extra point must never cost the headline line
                    extra[key] = {"error": repr(e)}
            res["extra"] = extra
            try:
                res["ber"] = ber_points(g)
            except Exception as e:
                res["ber"] = {"error": repr(e)}
            try:
                res["host_path"] = host_path(g, B=B, gen_device=local_rank)
            except Exception as e:
                res["host_path"] = {"error": repr(e)}
            try:
                res["coder_path"] = coder_path()
            except Exception as e:
                res["coder_path"] = {"error": repr(e)}
        if not args.no_extras:
            # the drop-in signature over a device list: ONE handle over every visible GPU, rank 0 only (the other
            # ranks have released their GPUs' memory and wait at the barrier below)
            if world > 1:
                dec.close()
                del y, out, gathered
                torch.cuda.empty_cache()
                dist.barrier(group=cpu_group)           # every rank has released its GPU's memory
            try:
                devs = list(range(torch.cuda.device_count())) if not rehearsal else [local_rank]
                res["host_path_devices"] = host_path(g, B=B, devices=devs, modes=("staged",), gen_device=local_rank)
            except Exception as e:
                res["host_path_devices"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    elif world > 1 and not args.no_extras:
        dec.close()
        del y, out, gathered
        torch.cuda.empty_cache()
        dist.barrier(group=cpu_group)
    dec.close()
    if world > 1:
        dist.barrier(group=cpu_group)                   # rank 0 may still be measuring host_path_devices: wait on the CPU
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
