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
                        "--batch", "256", "--backend", "gloo", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks"]["world_size"] == 2 and r["ranks"]["self_launched"] is True
    assert r["config"]["global_batch"] == 512 and r["value"] > 0 and r["scaling"] == "weak"
    assert r["ranks"]["ms_per_step_min"] <= r["ranks"]["ms_per_step_max"]
    assert 0 < r["roofline"]["frac"] <= 1.0 and r["roofline"]["hbm_probe_gbs"] > 1000
    # one record per rank: its step time, its dominant kernel's launch time, its GPU's copy rate
    pr = r["ranks"]["per_rank"]
    assert [p_["rank"] for p_ in pr] == [0, 1] and all(p_["avg_launch_ms"] > 0 and p_["hbm_probe_gbs"] > 1000 for p_ in pr)
    # the drop-in signature on a device-list handle, measured by rank 0 while rank 1 waits on the CPU
    hp = r["host_path_devices"]
    assert "error" not in hp, hp
    assert hp["staged"]["value"] > 0 and hp["locked_ranges_left"] == [0, 0] and hp["frames"] == 3 * 256


def test_bench_over_rccl_runs_where_two_gpus_are_visible(built):
    """`python bench.py --gpus 2` with the real backend (nccl = RCCL): needs two GPUs; on a one-GPU box the
    test is skipped, not failed (device_count() does not initialise the GPU)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL with two ranks needs two")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "512", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["ranks"]["backend"] == "nccl (RCCL)" and len(r["ranks"]["per_rank"]) == 2
    assert r["host_path_devices"]["devices"] == [0, 1]


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


@pytest.mark.parametrize("mode", ["staged", "lock_pages"])
def test_host_input_modes_give_the_same_bytes(built, mode):
    """ldpc_decode's two ways of moving pageable channel values -- through the library's pinned ring with
    the handle's own copy threads (default) and by page-locking the caller's pages for the call (opt-in) --
    for groups above and below the 4 MiB small-group limit, one and several groups, a tiny last group
    that shares a page with its predecessor, an unaligned buffer, a device list, host polling and
    asynchronous decodes.  Nothing may stay page-locked after any call, and the handle must close clean."""
    g, og, K, M, z = _graph(codes.RATE_1_2, 2304)
    big = np.empty(1301 * 2304 + 3, np.float32)
    y = big[3:]                                        # 12 bytes off the allocation's alignment
    y[:] = channel.awgn_frames(2304, 0, 1301, 0.8, seed=62).ravel()
    y = y.reshape(1301, 2304)
    want = None
    for devs, B, poll in ((None, 600, 0), (None, 600, 2), ([0, 0], 600, 0), (None, 1300, 2), ([0, 0, 0], 256, 2)):
        dec = L.Decoder(g, K, max_batch=B, algo="ms", max_iter=20, poll_interval=poll, devices=devs,
                        tune={"fused": False, "ldsp": False}, host_input=mode, host_copy_threads=3)
        for n in (1301, 1300, 601, 7):
            out, iters = dec.decode(y[:n])
            assert L.capi.host_locked_ranges() == (0, 0)
            if want is None:
                want = (out, iters)
                sample = oracle.decode(og, y[:48], "ms", max_iter=20)
                assert np.array_equal(out[:48 * K // 8], sample["out"]) and np.array_equal(iters[:48], sample["iters"])
            assert np.array_equal(out, want[0][:n * K // 8]) and np.array_equal(iters, want[1][:n]), (devs, B, poll, n)
        dec.close()                                     # raises if a locked block was left behind


def test_two_handles_decode_the_same_buffer_concurrently(built):
    """Two decoders on two Python threads decode the SAME pageable buffer at the same time, in both input
    modes (lock mode: the second call finds the pages on this library's record and stages instead)."""
    import threading
    g, og, K, M, z = _graph(codes.RATE_1_2, 2304)
    y = channel.awgn_frames(2304, 0, 1200, 0.8, seed=63)
    ref = L.Decoder(g, K, max_batch=600, algo="ms", max_iter=20, tune={"fused": False, "ldsp": False})
    want = ref.decode(y)
    ref.close()
    for mode in ("staged", "lock_pages"):
        decs = [L.Decoder(g, K, max_batch=600, algo="ms", max_iter=20, tune={"fused": False, "ldsp": False},
                          host_input=mode) for _ in range(2)]
        res = [None, None]

        def run(i):
            res[i] = [decs[i].decode(y) for _ in range(3)]
        ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for i in range(2):
            for out, iters in res[i]:
                assert np.array_equal(out, want[0]) and np.array_equal(iters, want[1]), mode
            decs[i].close()
        assert L.capi.host_locked_ranges() == (0, 0)


@pytest.mark.parametrize("threads", [1, 2, 16])
def test_staged_input_with_any_number_of_copy_threads(built, threads):
    """host_copy_threads = 1 (the stager alone, no helpers), 2 and 16 (the maximum): groups just above and just below the
    4 MiB small-group limit, a call without iteration counts, and an output buffer shorter than the decoded stream
    (Coder::decode writes srcLength bytes, MyLdpc.cpp:571-618)."""
    import ctypes
    g, og, K, M, z = _graph(codes.RATE_1_2, 2304)
    y = channel.awgn_frames(2304, 0, 1000, 0.8, seed=64)
    want = oracle.decode(og, y[:96], "ms", max_iter=20)
    ref = None
    for B in (455, 456, 1000):                       # 455 * 2304 * 4 = 4 193 280 B < 4 MiB < 456 * 2304 * 4
        dec = L.Decoder(g, K, max_batch=B, algo="ms", max_iter=20, tune={"fused": False, "ldsp": False},
                        host_copy_threads=threads)
        out, iters = dec.decode(y)
        if ref is None:
            ref = (out, iters)
            assert np.array_equal(out[:96 * K // 8], want["out"]) and np.array_equal(iters[:96], want["iters"])
        assert np.array_equal(out, ref[0]) and np.array_equal(iters, ref[1]), B
        out2, _ = dec.decode(y, want_iters=False)
        assert np.array_equal(out2, ref[0])
        short = np.full(1000, 0xEE, np.uint8)        # room for 6 frames and 136 bytes of the 7th; the rest stays untouched
        from myldpccppapi_amd import _lib
        _lib.check(_lib.load().ldpc_decode(dec._h, y.ctypes.data, 1000, short.ctypes.data, 1000, None))
        assert np.array_equal(short, ref[0][:1000])
        dec.close()


def test_host_path_randomized(built):
    """Seeded random walk over the host-buffer entry point: frame counts, launch-group sizes (groups above and below the
    4 MiB limit, one to many groups), both input modes, device lists, copy-thread counts, polling, unaligned input
    buffers -- every result against one reference decode of the whole stream; nothing page-locked is left behind."""
    g, og, K, M, z = _graph(codes.RATE_1_2, 2304)
    total = 3000
    big = np.empty(total * 2304 + 16, np.float32)
    ref_dec = L.Decoder(g, K, max_batch=total, algo="ms", max_iter=10, tune={"fused": False, "ldsp": False})
    y0 = channel.awgn_frames(2304, 0, total, 0.85, seed=65)
    want_out, want_it = ref_dec.decode(y0)
    ref_dec.close()
    sample = oracle.decode(og, y0[:32], "ms", max_iter=10)
    assert np.array_equal(want_out[:32 * K // 8], sample["out"]) and np.array_equal(want_it[:32], sample["iters"])
    rng = np.random.default_rng(2026)
    kb = K // 8
    for case in range(36):
        off = int(rng.integers(0, 16))                       # the buffer starts 4 * off bytes into its allocation
        y = big[off:off + total * 2304].reshape(total, 2304)
        y[:] = y0
        B = int(rng.choice([1, 7, 64, 300, 455, 456, 1000, 1500, 3000]))
        devs = [None, None, [0, 0], [0, 0, 0]][int(rng.integers(0, 4))]
        mode = ["staged", "lock_pages"][int(rng.integers(0, 2))]
        dec = L.Decoder(g, K, max_batch=B, algo="ms", max_iter=10, devices=devs, host_input=mode,
                        host_copy_threads=int(rng.integers(1, 7)), poll_interval=int(rng.choice([0, 2])),
                        tune={"fused": False, "ldsp": False})
        for _ in range(2):
            lo = int(rng.integers(0, total - 1))
            n = int(rng.integers(1, min(total - lo, 40 * B if B < 64 else total) + 1))
            out, iters = dec.decode(y[lo:lo + n])
            assert np.array_equal(out, want_out[lo * kb:(lo + n) * kb]), (case, B, devs, mode, lo, n)
            assert np.array_equal(iters, want_it[lo:lo + n]), (case, B, devs, mode, lo, n)
            assert L.capi.host_locked_ranges() == (0, 0)
        dec.close()


def test_cpp_coder_host_input_modes(built, tmp_path):
    """Coder::setHostInput: the reference's round trip (Test.cpp) with the caller's postCode staged through the library's
    pinned ring (default) and page-locked in place (opt-in), in several launch groups (1100 frames of 9216 B x 1024 per
    group = 9.4 MB groups): identical decoded bytes."""
    exe = str(tmp_path / "coder_roundtrip")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "coder_roundtrip.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "myldpccppapi_amd"), "-lmyldpc", "-lldpc_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "myldpccppapi_amd")])
    outs = []
    for mode in ("0", "1", "2"):
        pre = str(tmp_path / ("h" + mode))
        cmd = [exe, "0", "2304", str(1100 * 144), "1024", "2.6", "MS", "11", "--dump", pre, "--host-input", mode]
        out = subprocess.run(cmd, capture_output=True, text=True)
        assert out.returncode == 0 and "ParityFail=0" in out.stdout, out.stdout + out.stderr
        outs.append(open(pre + ".out", "rb").read())
    assert outs[0] == outs[1] == outs[2]
