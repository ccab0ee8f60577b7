"""Multi-GPU paths on the one-GPU test box: a device list behind one handle (two per-device decoders
sharing GPU 0), two ranks that run the HIP decoder on their frame shards (gloo rendezvous, both
on GPU 0), and the benchmark starting its own ranks.  The decoders are the product's
(libldpc_hip.so); the oracle only checks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
import myldpccppapi_amd as L
from myldpccppapi_amd import channel, codes
from util import converged_frames, wimax_oracle_graph

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph(rate, N):
    og, rows, cols, K, M, z = wimax_oracle_graph(rate, N)
    return L.Graph(rows, cols, M, N), og, K, M, z


@pytest.mark.parametrize("algo", ["sp", "ms", "layered"])
def test_device_list_handle_equals_single_device(built, algo):
    """devices = {0, 0}: ldpc_decode cuts a ragged 700-frame batch into two contiguous ranges, one
    host thread + stream + staging per entry; bytes and iteration counts must equal the single-device
    handle's (and the oracle's), also when each range spans several launch groups."""
    g, og, K, M, z = _graph(codes.RATE_1_2, 1152)
    B = 700
    y = channel.awgn_frames(1152, 0, B, 0.78, seed=51)
    want = oracle.decode(og, y, algo, max_iter=25, layer_rows=z)
    rows, cols = codes.wimax_edges(codes.RATE_1_2, 1152)
    n_conv = int(converged_frames(rows, cols, M, want["hard"]).sum())
    for max_batch, fpl in ((512, 0), (128, 2 if algo != "layered" else 0)):
        one = L.Decoder(g, K, max_batch=max_batch, algo=algo, max_iter=25, layer_rows=z, frames_per_lane=fpl)
        two = L.Decoder(g, K, max_batch=max_batch, algo=algo, max_iter=25, layer_rows=z, frames_per_lane=fpl,
                        devices=[0, 0])
        o1, i1 = one.decode(y)
        for _ in range(2):                                   # staging slots are reused by the second call
            o2, i2 = two.decode(y)
            assert np.array_equal(o2, o1) and np.array_equal(i2, i1), (algo, max_batch)
        assert np.array_equal(o1, want["out"]) and np.array_equal(i1, want["iters"])
        st = two.stats()
        if max_batch >= 350:                                 # one launch group per device: stats cover all frames
            assert st["frames"] == B and st["frames_converged"] == n_conv
            assert st["batch_time"] == int(want["iters"].max())
        # fewer frames than devices, and a device-pointer call on the list handle is refused
        o3, i3 = two.decode(y[:1])
        assert np.array_equal(o3, want["out"][:K // 8]) and i3[0] == want["iters"][0]
        with pytest.raises(L.LdpcError) as e:
            two.decode_device(1, 1, 1, 1)
        assert e.value.code == 5
        one.close()
        two.close()


@pytest.mark.parametrize("pack", [L.PACK_BYTES, L.PACK_BITS])
def test_device_list_with_frames_not_byte_aligned(built, pack):
    """(648, 324): K % 8 = 4, so a frame's first byte is (frame*K)/8 per launch group -- shard
    boundaries are chosen where that stays exact (ldpc_hip.hip: shard_unit), and three devices
    sharing the GPU must reproduce the single-device bytes in both packings."""
    g, og, K, M, z = _graph(codes.RATE_1_2, 648)
    B = 301
    y = channel.awgn_frames(648, 0, B, 0.7, seed=52)
    for max_batch in ((512, 64) if pack == L.PACK_BYTES else (512,)):
        one = L.Decoder(g, K, max_batch=max_batch, algo="ms", pack_mode=pack, frames_per_lane=1)
        many = L.Decoder(g, K, max_batch=max_batch, algo="ms", pack_mode=pack, frames_per_lane=1, devices=[0, 0, 0])
        o1, i1 = one.decode(y)
        o2, i2 = many.decode(y)
        assert np.array_equal(o2, o1) and np.array_equal(i2, i1), (pack, max_batch)
        if max_batch == 512:
            want = oracle.decode(og, y, "ms", pack_mode=pack)
            assert np.array_equal(o1, want["out"]) and np.array_equal(i1, want["iters"])
        one.close()
        many.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import myldpccppapi_amd as L
from myldpccppapi_amd import channel, codes, sharding
rank, world, total = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), {total}
dist.init_process_group("gloo", rank=rank, world_size=world)
rate, N = codes.RATE_2_3_B, 960
K, M, z = codes.wimax_dims(rate, N)
rows, cols = codes.wimax_edges(rate, N)
g = L.Graph(rows, cols, M, N)
dec = L.Decoder(g, K, max_batch=64, algo="sp", max_iter=30, device=0, frames_per_lane=2)    # the HIP decoder

def decode_fn(lo, hi):
    # this rank's frames of the shared counter-based channel, generated on its GPU
    y = channel.awgn_device(N, lo, hi - lo, 0.62, seed=77, device=0)
    out, iters = dec.decode(y.cpu().numpy())
    return torch.from_numpy(out.copy())

res = sharding.decode_sharded(decode_fn, total, K, dst={dst})
if res is not None:
    np.save(os.path.join({tmp!r}, "rank%d.npy" % rank), res.numpy())
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("total,dst", [(203, None), (130, 0)])
def test_two_ranks_run_the_hip_decoder_on_their_shards(built, tmp_path, total, dst):
    """Two processes (gloo rendezvous on 127.0.0.1, both on GPU 0) each decode their shard_range
    with L.Decoder and gather the bytes with sharding.gather_decoded: the result must be the
    single-process HIP decode of all frames, which must be the oracle's."""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(root=ROOT, total=total, dst=dst, tmp=str(tmp_path)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    rate, N = codes.RATE_2_3_B, 960
    g, og, K, M, z = _graph(rate, N)
    y = channel.awgn_device(N, 0, total, 0.62, seed=77, device=0).cpu().numpy()
    dec = L.Decoder(g, K, max_batch=256, algo="sp", max_iter=30)
    single, _ = dec.decode(y)
    dec.close()
    assert np.array_equal(single, oracle.decode(og, y, "sp", max_iter=30)["out"])
    for r in ((0, 1) if dst is None else (dst,)):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert np.array_equal(got, single), r
    if dst is not None:
        assert not (tmp_path / "rank1.npy").exists()


def test_bench_starts_its_own_ranks(built):
    """`python bench.py --gpus 2` invoked plainly (no launcher, no WORLD_SIZE): it must start two
    ranks itself, exit 0 and print one JSON line with n_gpus = 2.  --backend gloo lets both ranks
    share this box's one GPU (the collective runs on CPU copies; decoding is the HIP path)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "256", "--backend", "gloo", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks"]["world_size"] == 2 and r["ranks"]["self_launched"] is True
    assert r["config"]["global_batch"] == 512 and r["value"] > 0 and r["scaling"] == "weak"
    assert r["ranks"]["ms_per_step_min"] <= r["ranks"]["ms_per_step_max"]
    assert 0 < r["roofline"]["frac"] <= 1.0 and r["roofline"]["hbm_probe_gbs"] > 1000


def test_cpp_coder_over_a_device_list(built, tmp_path):
    """Coder::setDevices({0, 0}): the reference's round trip (Test.cpp) through a Coder that spans two
    device decoders; decoded bytes equal the single-device Coder's on the same noisy stream."""
    exe = str(tmp_path / "coder_roundtrip")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "coder_roundtrip.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "myldpccppapi_amd"), "-lmyldpc", "-lldpc_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "myldpccppapi_amd")])
    for mode in ("SP", "MS", "CPU", "TDMPCL"):
        outs = []
        for devs in (None, "0,0"):
            pre = str(tmp_path / ("d%s_%s" % (mode, "1" if devs is None else "2")))
            cmd = [exe, "0", "2304", "40000", "64", "2.6", mode, "9", "--dump", pre]
            if devs:
                cmd += ["--devices", devs]
            out = subprocess.run(cmd, capture_output=True, text=True)
            assert out.returncode == 0 and "ParityFail=0" in out.stdout, out.stdout + out.stderr
            outs.append(open(pre + ".out", "rb").read())
            assert open(pre + ".post", "rb").read() == open(str(tmp_path / ("d%s_1.post" % mode)), "rb").read()
        assert outs[0] == outs[1], mode


def test_caller_locked_input_is_copied_directly(built):
    """A caller that has page-locked its input itself (here: a pinned torch tensor) gets plain copies --
    the library locks nothing -- for one device and for a device list, in several launch groups."""
    import torch
    g, og, K, M, z = _graph(codes.RATE_3_4_A, 1152)
    B = 333
    y = channel.awgn_frames(1152, 0, B, 0.5, seed=57)
    yp = torch.from_numpy(y).pin_memory()
    want = oracle.decode(og, y, "ms", max_iter=20)
    for devs in (None, [0, 0]):
        dec = L.Decoder(g, K, max_batch=64, algo="ms", max_iter=20, frames_per_lane=1, devices=devs)
        for src in (yp.numpy(), y):
            out, iters = dec.decode(src)
            assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"]), devs
        dec.close()


@pytest.mark.parametrize("algo,poll", [("sp", 0), ("ms", 2), ("layered", 0)])
def test_streams_on_one_device_equal_the_single_stream_decoder(built, algo, poll):
    """cfg.streams = 2 / 3: sub-decoders of max_batch / streams frames on streams of their own (same GPU);
    device-pointer calls (asynchronous and polled) and the host-buffer call must give the single-stream
    decoder's bytes and iteration counts for full, ragged and tiny batches, and the caller's stream must
    see the result in stream order (the copies below are enqueued on it right after the call)."""
    import torch
    g, og, K, M, z = _graph(codes.RATE_1_2, 2304)
    B = 2048
    y = channel.awgn_device(2304, 0, B, 0.8, seed=61, device=0)
    yh = y.cpu().numpy()
    one = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=20, layer_rows=z, poll_interval=poll,
                    tune={"fused": False, "ldsp": False} if algo != "layered" else None)
    want = {}
    for n in (B, 1500, 300, 1):
        want[n] = one.decode(yh[:n])
    sample = oracle.decode(og, yh[:64], algo, max_iter=20, layer_rows=z)
    assert np.array_equal(want[B][0][:64 * K // 8], sample["out"]) and np.array_equal(want[B][1][:64], sample["iters"])
    one.close()
    for streams in (2, 3):
        dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=20, layer_rows=z, poll_interval=poll, streams=streams,
                        tune={"fused": False, "ldsp": False} if algo != "layered" else None)
        s = torch.cuda.Stream()
        for n in (B, 1500, 300, 1, B):
            out = torch.zeros(L.out_bytes(K, n), dtype=torch.uint8, device="cuda")
            it = torch.zeros(n, dtype=torch.int32, device="cuda")
            with torch.cuda.stream(s):
                dec.decode_device(y.data_ptr(), n, out.data_ptr(), out.numel(), it.data_ptr(), s.cuda_stream)
                o, i = out.cpu().numpy(), it.cpu().numpy()          # stream-ordered behind the call
            assert np.array_equal(o, want[n][0]) and np.array_equal(i, want[n][1]), (streams, n)
            st = dec.stats()
            assert st["frames"] == n and st["batch_time"] == int(want[n][1].max())
        oh, ih = dec.decode(yh[:1500])
        assert np.array_equal(oh, want[1500][0]) and np.array_equal(ih, want[1500][1]), streams
        dec.close()
    # too small a batch per stream, or frames that are not byte aligned: one plain decoder behind the handle
    small = L.Decoder(g, K, max_batch=600, algo=algo, max_iter=20, layer_rows=z, streams=2)
    o, i = small.decode(yh[:300])
    assert np.array_equal(o, want[300][0]) and np.array_equal(i, want[300][1])
    small.close()


def test_cpp_coder_with_streams(built, tmp_path):
    """Coder::setStreams(2): 1100 frames of the (2304, 1152) code in launch groups of 1024, each group cut
    into two ranges on streams of their own; decoded bytes equal the plain Coder's on the same noisy stream
    (host polling on: each range has a host thread)."""
    exe = str(tmp_path / "coder_roundtrip")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "coder_roundtrip.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "myldpccppapi_amd"), "-lmyldpc", "-lldpc_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "myldpccppapi_amd")])
    for mode in ("SP", "MS"):
        outs = []
        for streams in (0, 2):
            pre = str(tmp_path / ("s%s_%d" % (mode, streams)))
            cmd = [exe, "0", "2304", str(1100 * 144), "1024", "2.6", mode, "11", "--dump", pre]
            if streams:
                cmd += ["--streams", str(streams)]
            out = subprocess.run(cmd, capture_output=True, text=True)
            assert out.returncode == 0 and "ParityFail=0" in out.stdout, out.stdout + out.stderr
            outs.append(open(pre + ".out", "rb").read())
        assert outs[0] == outs[1], mode
