"""Parity tests proper: the HIP path (through the C ABI) against the golden vectors
made from the reference's kernels and against the CPU oracle, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import oracle
import myldpccppapi_amd as L
from myldpccppapi_amd import channel, codes
from util import converged_frames, golden_files, kernel_choice, load_golden, wimax_oracle_graph

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOOD = golden_files("flood")
LAYERED = golden_files("layered")
MSCL = golden_files("mscl")
TDMPHOST = golden_files("tdmphost")
f32 = np.float32


def _graph(rate, N):
    og, rows, cols, K, M, z = wimax_oracle_graph(rate, N)
    return L.Graph(rows, cols, M, N), og, K, M, z


@pytest.mark.parametrize("path", FLOOD, ids=lambda p: p.split("flood_")[-1][:-4])
@pytest.mark.parametrize("algo", ["ms", "sp"])
@pytest.mark.parametrize("V", [1, 2, 4])
def test_flooding_bit_exact_vs_reference_golden(built, path, algo, V):
    gd = load_golden(path)
    g, og, K, M, z = _graph(int(gd["rate"]), int(gd["N"]))
    y = gd["y"]
    B = y.shape[0]
    dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=int(gd["times"]), frames_per_lane=V)
    out, iters = dec.decode(y)
    assert np.array_equal(out, gd[algo + "_out"])                       # the reference's bytes
    st = dec.stats()
    assert st["batch_time"] == int(gd[algo + "_time"])                  # its "Time="
    assert st["frames_converged"] == int((gd[algo + "_flags"] == 0).sum())
    o = oracle.decode(og, y, algo, max_iter=int(gd["times"]))
    assert np.array_equal(iters, o["iters"])
    # all N hard bits, not only the K packed ones
    assert np.array_equal(dec.dump(3, B).astype(np.uint8), gd[algo + "_hard"])
    dec.close()


@pytest.mark.parametrize("path", FLOOD[:4] + FLOOD[-2:], ids=lambda p: p.split("flood_")[-1][:-4])
@pytest.mark.parametrize("algo", ["ms", "sp"])
def test_intermediate_messages_bit_exact(built, path, algo):
    """Messages after `tap_iter` rounds, compared bitwise (tolerance 0 ulp) with what the
    reference kernels hold, on frames still running at that round."""
    gd = load_golden(path)
    g, og, K, M, z = _graph(int(gd["rate"]), int(gd["N"]))
    y, tap = gd["y"], int(gd["tap_iter"])
    B = y.shape[0]
    o = oracle.decode(og, y, algo, max_iter=int(gd["times"]))
    dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=int(gd["times"]), frames_per_lane=2)
    dec.set_tap(tap)
    dec.decode(y)
    R, Q = dec.dump(0, B), dec.dump(1, B)
    run_r = np.nonzero(o["iters"] >= tap)[0]
    run_q = np.nonzero(o["iters"] > tap)[0]
    assert len(run_r) > 0
    if algo == "ms":
        assert np.array_equal(R[run_r], gd["ms_tap_r"][run_r], equal_nan=True)
        assert np.array_equal(Q[run_q], gd["ms_tap_q"][run_q], equal_nan=True)
    else:
        # stored d = r0 - r1 form: r0 = (1+d)/2, r1 = (1-d)/2 (decodeCL.c:39-40), d_q = q0 - q1 (:37)
        r0 = (f32(1) + R) * f32(0.5)
        r1 = (f32(1) - R) * f32(0.5)
        assert np.array_equal(r0[run_r], gd["sp_tap_r0"][run_r], equal_nan=True)
        assert np.array_equal(r1[run_r], gd["sp_tap_r1"][run_r], equal_nan=True)
        dq = gd["sp_tap_q0"] - gd["sp_tap_q1"]
        assert np.array_equal(Q[run_q], dq[run_q], equal_nan=True)
        # exp(8 y), decodeCL.c:9: bitwise against the host libm's expf on fl32(8 * y) -- the function
        # the oracle calls (ldpc_expf equals it on all 2^32 inputs, tests/test_host_cpu.py)
        import ctypes
        libm = ctypes.CDLL("libm.so.6")
        libm.expf.restype, libm.expf.argtypes = ctypes.c_float, [ctypes.c_float]
        arg = (f32(8.0) * y.astype(f32)).ravel()
        want = np.array([libm.expf(float(v)) for v in arg], f32)
        assert np.array_equal(dec.dump(2, B).ravel().view(np.uint32), want.view(np.uint32))
    dec.close()


@pytest.mark.parametrize("path", LAYERED, ids=lambda p: p.split("layered_")[-1][:-4])
@pytest.mark.parametrize("V", [1, 2, 4])
def test_layered_bit_exact_vs_fused_reference_kernel(built, path, V):
    gd = load_golden(path)
    g, og, K, M, z = _graph(int(gd["rate"]), int(gd["N"]))
    y = gd["y"]
    dec = L.Decoder(g, K, max_batch=y.shape[0], algo="layered", max_iter=int(gd["times"]),
                    layer_rows=z, frames_per_lane=V)
    out, iters = dec.decode(y)
    assert np.array_equal(out, gd["out"])
    o = oracle.decode(og, y, "layered", max_iter=int(gd["times"]), layer_rows=z, tap_iter=2)
    assert np.array_equal(iters, o["iters"])
    dec.set_tap(2)
    dec.decode(y)
    run = np.nonzero(o["iters"] >= 2)[0]
    assert np.array_equal(dec.dump(0, y.shape[0])[run], o["taps"]["r"][run])
    assert np.array_equal(dec.dump(2, y.shape[0])[run], o["taps"]["post"][run])
    dec.close()


@pytest.mark.parametrize("path", MSCL, ids=lambda p: p.split("mscl_")[-1][:-4])
def test_fused_flooding_bit_exact_vs_reference_kernel(built, path):
    """LDPC_ALGO_MS_FUSED (DecodeMSCL) against the reference's decodeOnceMS kernel outputs and
    the oracle's iteration counts / messages."""
    gd = load_golden(path)
    g, og, K, M, z = _graph(int(gd["rate"]), int(gd["N"]))
    y = gd["y"]
    B = y.shape[0]
    o = oracle.decode(og, y, "ms_fused", max_iter=120, tap_iter=2)
    # the record kernel (flood_ldsp_kernel, the default) and the LDS-resident fused_flood_kernel
    for ldsp in ("1", "0"):
        dec = L.Decoder(g, K, max_batch=B, algo="ms_fused", max_iter=120, layer_rows=z,
                        tune={"ldsp": ldsp == "1", "ldsp_grid": 3})
        out, iters = dec.decode(y)
        assert np.array_equal(out, gd["out"]), ldsp
        assert np.array_equal(iters, o["iters"]), ldsp
        dec.set_tap(2)
        dec.decode(y)
        run = np.nonzero(o["iters"] >= 2)[0]
        assert np.array_equal(dec.dump(0, B)[run], o["taps"]["r"][run]), ldsp
        assert np.array_equal(dec.dump(2, B)[run], o["taps"]["post"][run]), ldsp
        dec.close()


def test_layered_streaming_and_fused_paths_agree(built):
    """Short QC codes take the fused LDS-resident kernel (one launch); tune fused=False keeps
    the one-launch-per-layer streaming kernels.  Both must give the oracle's bits."""
    for rate, N, sigma, B in ((0, 576, 0.8, 70), (3, 1152, 0.55, 9), (0, 2304, 0.9, 6), (5, 960, 0.45, 130)):
        g, og, K, M, z = _graph(rate, N)
        y = channel.awgn_frames(N, 0, B, sigma, seed=13)
        want = oracle.decode(og, y, "layered", layer_rows=z, tap_iter=3)
        # "ldsp": posterior in LDS, 16-byte check records in cache (layered_ldsp_kernel), here with a
        # grid of 4 persistent workgroups so that each one walks over several frames
        for fused in ("1", "ldsp", "0"):
            dec = L.Decoder(g, K, max_batch=B, algo="layered", layer_rows=z, tune=kernel_choice(fused, 4))
            out, iters = dec.decode(y)
            assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"]), (rate, N, fused)
            assert dec.stats()["frames_converged"] == int((want["iters"] < 40).sum()) or want["iters"].max() == 40
            dec.set_tap(3)
            dec.decode(y)
            run = np.nonzero(want["iters"] >= 3)[0]
            assert np.array_equal(dec.dump(0, B)[run], want["taps"]["r"][run]), fused
            assert np.array_equal(dec.dump(2, B)[run], want["taps"]["post"][run]), fused
            dec.close()


def test_min_sum_fused_and_streaming_paths_agree(built):
    """DecodeMS / DecodeCPU on short QC codes (circulant size given as layer_rows) run the MS
    chain's arithmetic in one LDS-resident launch; tune fused=False keeps the streaming
    kernels.  Both must equal the oracle (bytes, iteration counts, messages)."""
    for rate, N, sigma, B in ((4, 576, 0.55, 70), (0, 648, 0.8, 9), (3, 1152, 0.6, 9), (0, 2304, 0.9, 6), (5, 960, 0.45, 40)):
        g, og, K, M, z = _graph(rate, N)
        y = channel.awgn_frames(N, 0, B, sigma, seed=15)
        want = oracle.decode(og, y, "ms", tap_iter=2)
        # "ldsp": posteriors in LDS, one 16-byte record per check row (flood_ldsp_kernel), a grid of 4
        # persistent workgroups so that each one walks over several frames
        for fused in ("1", "ldsp", "0"):
            dec = L.Decoder(g, K, max_batch=B, algo="ms", layer_rows=z, tune=kernel_choice(fused, 4))
            out, iters = dec.decode(y)
            assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"]), (rate, N, fused)
            assert dec.stats()["batch_time"] == int(want["iters"].max())
            dec.set_tap(2)
            dec.decode(y)
            run = np.nonzero(want["iters"] >= 2)[0]
            assert np.array_equal(dec.dump(0, B)[run], want["taps"]["r"][run]), fused
            dec.close()
        if K % 8 == 0:
            dec = L.Decoder(g, K, max_batch=B, algo="ms", layer_rows=z, pack_mode=L.PACK_BITS)   # DecodeCPU
            out, iters = dec.decode(y)
            assert np.array_equal(out, oracle.decode(og, y, "ms", pack_mode=1)["out"])
            dec.close()


def test_sum_product_fused_and_streaming_paths_agree(built):
    """DecodeSP on short QC codes: the probability-domain sum-product in one LDS-resident launch
    (fused_sp_kernel) against the streaming kernels and the oracle, messages included."""
    for rate, N, sigma, B in ((4, 576, 0.5, 70), (0, 648, 0.75, 9), (0, 1152, 0.85, 9), (5, 960, 0.42, 40), (1, 672, 0.6, 12)):
        g, og, K, M, z = _graph(rate, N)
        y = channel.awgn_frames(N, 0, B, sigma, seed=16)
        want = oracle.decode(og, y, "sp", tap_iter=2)
        for fused in ("1", "0"):
            dec = L.Decoder(g, K, max_batch=B, algo="sp", layer_rows=z, tune={"fused": fused == "1"})
            out, iters = dec.decode(y)
            assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"]), (rate, N, fused)
            assert np.array_equal(dec.decode(y[:1])[0], oracle.decode(og, y[:1], "sp")["out"])
            dec.set_tap(2)
            dec.decode(y)
            run_r = np.nonzero(want["iters"] >= 2)[0]
            run_q = np.nonzero(want["iters"] > 2)[0]
            R, Q = dec.dump(0, B), dec.dump(1, B)
            assert np.array_equal(((f32(1) + R) * f32(0.5))[run_r], want["taps"]["r0"][run_r], equal_nan=True), fused
            assert np.array_equal(((f32(1) - R) * f32(0.5))[run_r], want["taps"]["r1"][run_r], equal_nan=True), fused
            dq = want["taps"]["q0"] - want["taps"]["q1"]
            assert np.array_equal(Q[run_q], dq[run_q], equal_nan=True), fused
            dec.close()


@pytest.mark.parametrize("algo", ["ms", "sp", "layered"])
def test_ragged_batches_and_chunking(built, algo):
    """frames not a multiple of the tile, more frames than max_batch (Coder::decode's
    chunk loop, MyLdpc.cpp:577-616), a single frame, zero frames."""
    rate, N = codes.RATE_2_3_B, 960
    g, og, K, M, z = _graph(rate, N)
    y = channel.awgn_frames(N, 0, 150, 0.66, seed=3)
    want = oracle.decode(og, y, algo, layer_rows=z)
    for max_batch, V in ((150, 1), (64, 1), (70, 2), (33, 4), (256, 4)):
        dec = L.Decoder(g, K, max_batch=max_batch, algo=algo, layer_rows=z, frames_per_lane=V)
        out, iters = dec.decode(y)
        assert np.array_equal(out, want["out"]), (max_batch, V)
        assert np.array_equal(iters, want["iters"])
        out1, it1 = dec.decode(y[5:6])
        assert np.array_equal(out1, oracle.decode(og, y[5:6], algo, layer_rows=z)["out"])
        out0, _ = dec.decode(np.zeros((0, N), np.float32))
        assert out0.size == 0
        dec.close()


@pytest.mark.parametrize("algo", ["ms", "sp", "layered"])
def test_k_not_byte_aligned_and_bit_packing(built, algo):
    """(648, 324): K/8 = 40.5.  toChar packs 40 whole bytes at (b*K)/8 (decodeCL.c:191-192);
    decodeCPU packs bit b*K+i (MyLdpc.cpp:765-774)."""
    g, og, K, M, z = _graph(codes.RATE_1_2, 648)
    y = channel.awgn_frames(648, 0, 11, 0.8, seed=4)
    for mode in (L.PACK_BYTES, L.PACK_BITS):
        dec = L.Decoder(g, K, max_batch=16, algo=algo, layer_rows=z, pack_mode=mode)
        out, iters = dec.decode(y)
        want = oracle.decode(og, y, algo, layer_rows=z, pack_mode=mode)
        assert np.array_equal(out, want["out"]), mode
        dec.close()


def test_configs0_as_written_one_frame_20_iterations_decodecpu_packing(built, tmp_path):
    """BASELINE.json configs[0] as written: the (648, 324) rate-1/2 code, batch = 1, 20 iterations, the
    reference's CPU decoder (`DecodeCPU`, MyLdpc.cpp:684-784: min-sum, bit-offset packing).  Through the C ABI
    (every kernel choice a one-frame decoder can make) and through the C++ class's DecodeCPU mode, against
    the oracle's restatement of decodeCPU; one frame at a time over a few noise levels, incl. one that does
    not converge in 20 iterations."""
    g, og, K, M, z = _graph(codes.RATE_1_2, 648)
    for sigma, seed in ((0.6, 1), (0.75, 2), (0.82, 3), (1.1, 4)):
        y = channel.awgn_frames(648, 0, 1, sigma, seed=seed)
        want = oracle.decode(og, y, "ms", max_iter=20, pack_mode=1)
        assert want["out"].size == 41                                  # (1 * 324 + 7) // 8 bytes
        for tune in (None, kernel_choice("1"), kernel_choice("ldsp"), kernel_choice("0")):
            dec = L.Decoder(g, K, max_batch=1, algo="ms", max_iter=20, layer_rows=z, pack_mode=L.PACK_BITS, tune=tune)
            out, iters = dec.decode(y)
            assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"]), (sigma, tune)
            assert dec.stats()["batch_time"] == int(want["iters"][0])
            dec.close()
    assert want["iters"][0] == 20                                      # the last point ran all 20 iterations
    # the class: Coder(324, 648, rate_1_2), one frame, setMaxIterations(20), DecodeCPU; its own encoder and test
    # channel (libc rand), dumped so that the oracle decodes exactly the floats the class decoded
    exe = str(tmp_path / "coder_roundtrip")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "coder_roundtrip.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "myldpccppapi_amd"), "-lmyldpc", "-lldpc_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "myldpccppapi_amd")])
    pre = str(tmp_path / "c0")
    p = subprocess.run([exe, "0", "648", "40", "1", "2.5", "CPU", "5", "--iters", "20", "--dump", pre],
                       capture_output=True, text=True)
    assert p.returncode == 0 and "ParityFail=0" in p.stdout, p.stdout + p.stderr
    post = np.fromfile(pre + ".post", np.float32).reshape(1, 648)
    o = oracle.decode(og, post, "ms", max_iter=20, pack_mode=1)
    got = np.fromfile(pre + ".out", np.uint8)
    assert np.array_equal(got, o["out"][:40])                          # srcLength = 40 bytes are written
    assert ("Time=%d" % int(o["iters"][0])) in p.stdout


@pytest.mark.parametrize("algo", ["ms", "sp", "layered", "ms_fused"])
def test_degenerate_channel_values(built, algo):
    """Erasures (y = 0), huge values (exp overflows to inf -> inf/inf = NaN in the SP priors),
    infinities, NaNs, denormals and exact ties: whatever the reference's arithmetic makes of
    them, the HIP path must make the same of them."""
    rate, N = codes.RATE_1_2, 960
    g, og, K, M, z = _graph(rate, N)
    rng = np.random.default_rng(21)
    y = channel.awgn_frames(N, 0, 24, 0.7, seed=21)
    y[0, rng.choice(N, 200, replace=False)] = 0.0                 # erasures
    y[1, :] = 0.0                                                 # everything erased
    y[2, rng.choice(N, 50, replace=False)] = 12.0                 # exp(96) = inf
    y[3, rng.choice(N, 50, replace=False)] = -30.0
    y[4, rng.choice(N, 20, replace=False)] = np.inf
    y[5, rng.choice(N, 20, replace=False)] = -np.inf
    y[6, rng.choice(N, 5, replace=False)] = np.nan
    y[7, rng.choice(N, 100, replace=False)] = 1e-41               # denormal
    y[8, :] = 1.0                                                 # noiseless, all ties in min-sum
    y[9, :] = -1.0
    y[10, rng.choice(N, 300, replace=False)] *= 400.0             # magnitudes beyond the 1000 clip
    y[11, ::2] = 0.0
    for fused in (("1", "ldsp") if algo == "ms_fused" else (("1", "ldsp", "0") if algo in ("layered", "ms") else ("1", "0"))):
        dec = L.Decoder(g, K, max_batch=24, algo=algo, max_iter=15, layer_rows=z, tune=kernel_choice(fused))
        out, iters = dec.decode(y)
        want = oracle.decode(og, y, algo, max_iter=15, layer_rows=z)
        ok = np.ones(24, bool)
        if "undefined" in want:           # rows with every |Q| > 1000: undefined in the reference kernel
            ok = want["undefined"] == 0
        kb = K // 8
        assert np.array_equal(out.reshape(24, kb)[ok], want["out"].reshape(24, kb)[ok]), (algo, fused)
        assert np.array_equal(iters[ok], want["iters"][ok]), (algo, fused)
        if algo == "layered":
            # messages and posteriors after two iterations, bit patterns (signs of zeros included);
            # only where the oracle has a NaN the payload is left open
            tap = oracle.decode(og, y, algo, max_iter=15, layer_rows=z, tap_iter=2)
            dec.set_tap(2)
            dec.decode(y)
            run = ok & (want["iters"] >= 2)
            for which, name in ((0, "r"), (2, "post")):
                got, ref = dec.dump(which, 24)[run], tap["taps"][name][run]
                num = ~np.isnan(ref)
                assert np.array_equal(got.view(np.uint32)[num], ref.view(np.uint32)[num]), (fused, name)
                assert np.isnan(got[~num]).all(), (fused, name)
        dec.close()


def test_parameters_follow_the_oracle(built):
    g, og, K, M, z = _graph(codes.RATE_1_2, 1152)
    y = channel.awgn_frames(1152, 0, 40, 0.85, seed=6)
    for algo, kw in (("sp", dict(max_iter=20, llr_scale=4.0)), ("sp", dict(max_iter=7, llr_scale=8.0)),
                     ("ms", dict(max_iter=3)), ("layered", dict(max_iter=5)), ("ms", dict(max_iter=1))):
        dec = L.Decoder(g, K, max_batch=40, algo=algo, layer_rows=z, **kw)
        out, iters = dec.decode(y)
        want = oracle.decode(og, y, algo, layer_rows=z, **kw)
        assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"]), (algo, kw)
        dec.close()


def test_early_termination_switch_and_polling(built):
    g, og, K, M, z = _graph(codes.RATE_3_4_B, 576)
    y = channel.awgn_frames(576, 0, 100, 0.45, seed=8)            # everything converges quickly
    want = oracle.decode(og, y, "sp")
    dec = L.Decoder(g, K, max_batch=128, algo="sp", poll_interval=1)
    out, iters = dec.decode(y)
    assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"])
    st = dec.stats()
    assert st["iterations_launched"] == int(want["iters"].max()) < 40   # host stopped early
    assert st["frames_converged"] == 100
    dec.close()
    # early_term off: all rounds run, frames are not frozen; noisy frames (never clean) are
    # unaffected by the switch
    yn = channel.awgn_frames(576, 0, 20, 1.3, seed=9)
    wn = oracle.decode(og, yn, "ms")
    assert wn["iters"].min() == 40
    dec = L.Decoder(g, K, max_batch=32, algo="ms", early_term=False)
    out, iters = dec.decode(yn)
    assert np.array_equal(out, wn["out"])
    assert dec.stats()["iterations_launched"] == 40
    dec.close()


@pytest.mark.parametrize("V", [1, 2, 4])
def test_fp16_messages_follow_their_definition(built, V):
    """msg_dtype = f16 (this repository's extension; the reference is fp32 only): channel
    values and variable->check messages are rounded to binary16 when stored, arithmetic is
    fp32.  The HIP path must equal that definition (oracle msg_f16) bit for bit."""
    for rate, N, sigma, B in ((0, 648, 0.8, 9), (4, 576, 0.5, 70), (5, 1152, 0.42, 20), (0, 2304, 0.9, 5)):
        g, og, K, M, z = _graph(rate, N)
        y = channel.awgn_frames(N, 0, B, sigma, seed=12)
        dec = L.Decoder(g, K, max_batch=B, algo="ms", msg_dtype="f16", frames_per_lane=V)
        out, iters = dec.decode(y)
        want = oracle.decode(og, y, "ms", msg_f16=True, tap_iter=2)
        assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"]), (rate, N)
        dec.set_tap(2)
        dec.decode(y)
        run_r = np.nonzero(want["iters"] >= 2)[0]
        run_q = np.nonzero(want["iters"] > 2)[0]
        assert np.array_equal(dec.dump(0, B)[run_r], want["taps"]["r"][run_r])
        assert np.array_equal(dec.dump(1, B)[run_q], want["taps"]["q"][run_q])
        assert np.array_equal(dec.dump(2, B), y.astype(np.float16).astype(np.float32))
        dec.close()
    with pytest.raises(L.LdpcError) as e:
        L.Decoder(g, K, 8, algo="sp", msg_dtype="f16")
    assert e.value.code == 4


@pytest.mark.parametrize("algo", ["sp", "ms"])
def test_column_local_fusion_on_staircase_code(built, algo):
    """IRA (DVB-S2-profile) codes: the check kernel applies the variable-node update of the
    staircase parity columns itself (check_link_kernel).  Small instance (N = 12960) checked
    against the oracle: bytes, iteration counts and the messages of one round; fusion off
    (tune link_rows = -1) must give the same."""
    N2, K2 = 12960, 6480
    rows, cols = codes.dvbs2_profile_edges(N2, K2)
    g = L.Graph(rows, cols, N2 - K2, N2)
    og = oracle.Graph(rows, cols, N2 - K2, N2, K2)
    y = channel.awgn_frames(N2, 0, 70, 0.72 if algo == "ms" else 0.8, seed=14)
    want = oracle.decode(og, y, algo, max_iter=25, tap_iter=2)
    for rpw, variant in ((8, {}), (3, {}), (-1, {}), (16, {"link_deep": True}), (7, {"link_deep": True}), (2, {"link_deep": True}),
                         (8, {"link_narrow": False}), (5, {"link_half": True}), (16, {"link_half": True}),
                         # guided row chunks (the launch ends with chunks of a quarter of the rows), every kernel form
                         (16, {"link_guided": True}), (9, {"link_guided": True, "link_narrow": False}),
                         (16, {"link_guided": True, "link_half": True}), (12, {"link_guided": True, "link_deep": True}),
                         (16, {"link_guided": True, "tiles_first": False}), (8, {"tiles_first": True}),
                         # Q in edge order (as R) instead of the order its writers produce it, every kernel form; no fusion
                         (16, {"q_order": -1}), (6, {"q_order": -1, "link_narrow": False}), (16, {"q_order": -1, "link_half": True}),
                         (5, {"q_order": -1, "link_deep": True}), (-1, {"q_order": -1}), (16, {"q_order": 1, "merge": False})):
        for V in (1, 4):
            # link_deep: inputs requested two rows ahead; link_narrow False: V values per lane; link_half: 2 per lane (V = 4)
            dec = L.Decoder(g, K2, max_batch=70, algo=algo, max_iter=25, frames_per_lane=V,
                            tune=dict(variant, link_rows=rpw))
            out, iters = dec.decode(y)
            assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"]), (rpw, variant, V)
            dec.set_tap(2)
            dec.decode(y)
            run_q = np.nonzero(want["iters"] > 2)[0]
            Q = dec.dump(1, 70)
            if algo == "ms":
                assert np.array_equal(Q[run_q], want["taps"]["q"][run_q], equal_nan=True), (rpw, variant, V)
                assert np.array_equal(dec.dump(0, 70)[run_q], want["taps"]["r"][run_q]), (rpw, variant, V)
            else:
                dq = want["taps"]["q0"] - want["taps"]["q1"]
                assert np.array_equal(Q[run_q], dq[run_q], equal_nan=True), (rpw, variant, V)
            dec.close()


def test_generic_degree_kernels(built):
    """Degrees above the unrolled range take the generic kernels: the rate-5/6 seed has
    row weight 20 (flooding check / layered rows)."""
    g, og, K, M, z = _graph(codes.RATE_5_6, 1152)
    assert g.info()["max_row_deg"] == 20
    y = channel.awgn_frames(1152, 0, 30, 0.42, seed=10)
    for algo in ("ms", "sp"):
        dec = L.Decoder(g, K, max_batch=32, algo=algo)
        out, iters = dec.decode(y)
        want = oracle.decode(og, y, algo)
        assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"])
        dec.close()


def test_device_buffers_and_misuse(built):
    import torch
    g, og, K, M, z = _graph(codes.RATE_1_2, 576)
    y = channel.awgn_frames(576, 0, 48, 0.8, seed=11)
    dec = L.Decoder(g, K, max_batch=48, algo="ms")
    yd = torch.from_numpy(y).cuda()
    out = torch.empty(L.out_bytes(K, 48), dtype=torch.uint8, device="cuda")
    it = torch.empty(48, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        dec.decode_device(yd.data_ptr(), 48, out.data_ptr(), out.numel(), it.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    want = oracle.decode(og, y, "ms")
    assert np.array_equal(out.cpu().numpy(), want["out"]) and np.array_equal(it.cpu().numpy(), want["iters"])
    with pytest.raises(L.LdpcError) as e:
        dec.decode_device(yd.data_ptr(), 49, out.data_ptr(), out.numel())     # more than max_batch
    assert e.value.code == 1
    with pytest.raises(L.LdpcError):
        dec.decode_device(None, 4, out.data_ptr(), out.numel())
    dec.close()
    with pytest.raises(L.LdpcError):
        L.Decoder(g, K, 8, algo="layered", layer_rows=7)                       # does not divide M
    with pytest.raises(L.LdpcError):
        L.Decoder(g, K, 8, algo="ms", device=99)


def test_two_decoders_on_two_host_threads(built):
    """A handle is not re-entrant, but different handles may run concurrently from different host
    threads (thread-local error state, no shared mutable statics): each thread's results must be
    those of a serial run."""
    import threading
    g1, og1, K1, M1, z1 = _graph(codes.RATE_1_2, 1152)
    g2, og2, K2, M2, z2 = _graph(codes.RATE_3_4_A, 960)
    y1 = channel.awgn_frames(1152, 0, 300, 0.85, seed=17)
    y2 = channel.awgn_frames(960, 0, 300, 0.6, seed=18)
    w1 = oracle.decode(og1, y1, "sp")
    w2 = oracle.decode(og2, y2, "layered", layer_rows=z2)
    d1 = L.Decoder(g1, K1, max_batch=128, algo="sp", frames_per_lane=2)          # streaming, 3 groups
    d2 = L.Decoder(g2, K2, max_batch=300, algo="layered", layer_rows=z2)         # fused
    res = {}

    def work(name, dec, y):
        for _ in range(4):
            res[name] = dec.decode(y)

    ts = [threading.Thread(target=work, args=("a", d1, y1)), threading.Thread(target=work, args=("b", d2, y2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert np.array_equal(res["a"][0], w1["out"]) and np.array_equal(res["a"][1], w1["iters"])
    assert np.array_equal(res["b"][0], w2["out"]) and np.array_equal(res["b"][1], w2["iters"])
    d1.close()
    d2.close()


def test_cpp_coder_round_trip_like_test_cpp(built, tmp_path):
    """The reference's own pass signal (Test.cpp:105-110): ErrNum=0 after encode -> AWGN ->
    decode, for every decode mode of the C++ Coder class."""
    exe = str(tmp_path / "coder_roundtrip")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "coder_roundtrip.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "myldpccppapi_amd"), "-lmyldpc", "-lldpc_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "myldpccppapi_amd")])
    for mode in ("SP", "MS", "CPU", "TDMP", "TDMPCL", "MSCL"):
        # Test.cpp's code: z = 24, rate 3/4B; 10 kB payload, batch 64, 6 dB
        out = subprocess.run([exe, "4", "576", "10000", "64", "6", mode], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "ParityFail=0" in out.stdout and "ErrNum=0" in out.stdout, out.stdout
    out = subprocess.run([exe, "0", "2304", "5000", "16", "3.5", "SP"], capture_output=True, text=True)
    assert "ErrNum=0" in out.stdout, out.stdout
    # MyTest: the reference's CLI itself (same argv and printed fields, Test.cpp:16,35,48,56,61-112)
    mytest = os.path.join(ROOT, "myldpccppapi_amd", "MyTest")
    for mode in ("SP", "MS", "CPU", "TDMP", "TDMPCL", "MSCL"):
        out = subprocess.run([mytest, "20000", "128", "6", mode], capture_output=True, text=True,
                             env=dict(os.environ, MYTEST_SEED="5"))
        assert out.returncode == 0, out.stdout + out.stderr
        fields = dict(l.split("=", 1) for l in out.stdout.split() if "=" in l)
        assert fields["ErrNum"] == "0" and float(fields["ThroughPut"]) > 0 and "sd" in fields and "Time" in fields
        assert any(l.startswith(mode + ":") for l in out.stdout.split())
    assert subprocess.run([mytest, "1"], capture_output=True).returncode == 2


def test_device_channel_is_the_host_generator_bit_for_bit(built, tmp_path):
    """ldpc_awgn_device (Coder::test on the GPU, counter-based noise) against the same header
    evaluated on the host: every float identical, with and without code bits, for N % 4 != 0, for a
    frame range that starts in the middle, and split ranges concatenate to the whole."""
    import ctypes
    import torch
    from util import host_channel_lib
    lib = host_channel_lib(tmp_path)
    rng = np.random.default_rng(5)
    for N, frames, first, sd in ((648, 70, 0, 0.8), (67, 33, 1000, 0.5), (64800, 3, (1 << 33) + 5, 0.95), (2304, 9, 7, 0.0)):
        bits = rng.integers(0, 2, (frames, N)).astype(np.uint8)
        for b in (None, bits):
            want = np.empty((frames, N), np.float32)
            lib.awgn(want.ctypes.data_as(ctypes.c_void_p), frames, N,
                     None if b is None else b.ctypes.data_as(ctypes.c_void_p), sd, 99, first)
            got = channel.awgn_device(N, first, frames, sd, seed=99,
                                      codewords=None if b is None else torch.from_numpy(b).cuda())
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), (N, frames, b is None)
        whole = channel.awgn_device(N, first, frames, sd, seed=99)
        k = frames // 3
        parts = torch.cat([channel.awgn_device(N, first, k, sd, seed=99), channel.awgn_device(N, first + k, frames - k, sd, seed=99)])
        assert torch.equal(whole, parts)
    z = (channel.awgn_device(64800, 0, 64, 1.0, seed=1) - 1.0).double()
    assert abs(z.mean().item()) < 4 / np.sqrt(z.numel()) and abs(z.var().item() - 1) < 0.005


def test_device_error_count(built):
    import torch
    rng = np.random.default_rng(8)
    frames, per = 37, 405
    a = rng.integers(0, 256, (frames, per)).astype(np.uint8)
    b = a.copy()
    hit = rng.choice(frames * per, 300, replace=False)
    b.reshape(-1)[hit] ^= rng.integers(1, 256, 300).astype(np.uint8)
    b[5] = a[5]
    x = a ^ b
    want = (int(np.unpackbits(x).sum()), int((x != 0).sum()), int((x != 0).any(axis=1).sum()))
    assert channel.count_errors_device(torch.from_numpy(b).cuda(), torch.from_numpy(a).cuda(), frames) == want
    assert channel.count_errors_device(torch.from_numpy(x).cuda(), None, frames) == want
    assert channel.count_errors_device(torch.zeros(frames * per, dtype=torch.uint8, device="cuda"), None, frames) == (0, 0, 0)



def test_per_launch_timing_hooks(built):
    """set_timing: HIP events around every launch of the timed calls; 1 = every call, k = every k-th
    call (what bench.py uses inside its timed region), 0 = off.  Results do not depend on it."""
    g, og, K, M, z = _graph(codes.RATE_1_2, 2304)
    y = channel.awgn_frames(2304, 0, 130, 0.95, seed=31)
    want = oracle.decode(og, y, "sp", max_iter=12)
    dec = L.Decoder(g, K, max_batch=130, algo="sp", max_iter=12, early_term=False)
    out, iters = dec.decode(y)
    assert np.array_equal(out, want["out"])
    assert dec.kernel_times() == []
    dec.set_timing(True)
    for _ in range(2):
        out, iters = dec.decode(y)
    assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"])
    kt = dec.kernel_times()
    checks = [k for k in kt if k["phase"] == 0]
    varn = [k for k in kt if k["phase"] == 1]
    assert checks and varn and all(k["launches"] == 2 * 12 for k in checks + varn)
    assert all(k["ms_total"] > 0 and k["bytes_total"] > 0 and "kernel<sp," in k["name"] for k in checks + varn)
    # algorithmic bytes of the message kernels: 16 E + 4 N per frame-iteration (the last round's
    # variable phase writes no Q: a little less)
    total = sum(k["bytes_total"] for k in checks + varn)
    full = 2 * 130 * (16 * g.E + 4 * 2304) * 12
    assert 0.95 * full < total <= full
    st = dec.stats()
    assert st["ms_check"] > 0 and st["ms_var"] > 0 and st["launches_check"] == sum(k["launches"] for k in checks)
    dec.set_timing(2)                      # calls 1 and 3 of the next four are timed
    for _ in range(4):
        dec.decode(y)
    assert all(k["launches"] == 2 * 12 for k in dec.kernel_times() if k["phase"] in (0, 1))
    dec.set_timing(False)
    dec.decode(y)
    assert dec.kernel_times() == []
    dec.close()


def test_layered_on_random_quasi_cyclic_codes(built):
    """Random QC structures (circulant sizes that are / are not multiples of 64, single-layer
    columns in every position, rows of weight 1 to 24, one layer only, information parts that end
    inside a block column): the three layered implementations -- LDS-resident, LDS posterior +
    cached check records, one launch per layer -- against the oracle, bytes and iteration counts."""
    rng = np.random.default_rng(20261004)
    cases = []
    for z, mb, nb, dmax in ((5, 3, 9, 5), (17, 6, 14, 6), (64, 4, 12, 8), (65, 5, 30, 24), (100, 1, 10, 10),
                            (40, 8, 12, 3), (96, 12, 36, 7), (130, 3, 26, 24), (24, 10, 16, 2), (33, 7, 40, 20)):
        base = -np.ones((mb, nb), np.int64)
        for i in range(mb):
            d = int(rng.integers(1, min(dmax, nb) + 1))
            base[i, rng.choice(nb, d, replace=False)] = rng.integers(0, z, d)
        for j in range(nb):                       # every column is checked at least once
            if (base[:, j] < 0).all():
                i = int(rng.integers(0, mb))
                if (base[i] >= 0).sum() < 24:
                    base[i, j] = rng.integers(0, z)
        keep = [j for j in range(nb) if (base[:, j] >= 0).any()]
        base = base[:, keep]
        nb = len(keep)
        K = max(8, ((nb - min(mb, nb - 1)) * z - int(rng.integers(0, z))) // 8 * 8)
        cases.append((z, base, min(K, nb * z // 8 * 8)))
    for z, base, K in cases:
        mb, nb = base.shape
        rows, cols = codes.qc_edges(base, z)
        M, N = mb * z, nb * z
        g = L.Graph(rows, cols, M, N)
        og = oracle.Graph(rows, cols, M, N, K)
        B = 11
        sigma = 0.9 if M * 2 > N else 0.6
        y = channel.awgn_frames(N, 0, B, sigma, seed=z)
        y[3, ::3] = 0.0                           # zeros: rows that take the slow path of the record kernel
        want = oracle.decode(og, y, "layered", layer_rows=z, max_iter=9)
        ok = want["undefined"] == 0 if "undefined" in want else np.ones(B, bool)
        kb = K // 8
        for mode in ("fused", "ldsp", "stream"):
            tune = kernel_choice({"fused": "1", "ldsp": "ldsp", "stream": "0"}[mode], 3)
            dec = L.Decoder(g, K, max_batch=B, algo="layered", layer_rows=z, max_iter=9, tune=tune)
            out, iters = dec.decode(y)
            assert np.array_equal(out.reshape(B, kb)[ok], want["out"].reshape(B, kb)[ok]), (z, base.shape, K, mode)
            assert np.array_equal(iters[ok], want["iters"][ok]), (z, base.shape, K, mode)
            dec.close()
            # flooding min-sum on the same structures (LDS-resident / record kernel / streaming)
            if mode == "fused":
                want_ms = oracle.decode(og, y, "ms", max_iter=9)
            dec = L.Decoder(g, K, max_batch=B, algo="ms", layer_rows=z, max_iter=9, tune=tune)
            out, iters = dec.decode(y)
            assert np.array_equal(out, want_ms["out"]) and np.array_equal(iters, want_ms["iters"]), (z, base.shape, K, mode, "ms")
            dec.close()


@pytest.mark.parametrize("algo,f16", [("sp", False), ("ms", False), ("ms", True)])
def test_tail_compaction_of_running_frames(built, algo, f16):
    """Early termination with host polling: once at most `threshold` frames of a multi-tile batch are
    still running, their state moves into a one-tile child decoder that finishes them.  Bits,
    iteration counts and the converged count must be those of the oracle (frames are independent);
    with compaction off the same."""
    g, og, K, M, z = _graph(codes.RATE_1_2, 1152)
    B = 700
    rng = np.random.default_rng(40)
    y = channel.awgn_frames(1152, 0, B, 0.62, seed=41)            # most frames converge within a few rounds ...
    hard = rng.choice(B, 90, replace=False)
    y[hard] = channel.awgn_frames(1152, 5000, 90, 1.05, seed=42)  # ... some scattered ones (almost) never do: two child tiles
    want = oracle.decode(og, y, algo, max_iter=30, msg_f16=f16)
    assert (want["iters"] == 30).sum() >= 65 and (want["iters"] < 12).sum() > B - 120
    rows, cols = codes.wimax_edges(codes.RATE_1_2, 1152)
    n_conv = int(converged_frames(rows, cols, M, want["hard"]).sum())     # syndrome-clean, not iters < max
    assert n_conv >= int((want["iters"] < 30).sum())
    for compact, q_order in ((512, 0), (7, 0), (-1, 0), (512, -1)):      # q_order -1: Q in edge order (parent and child)
        for fpl in (1, 2, 4):
            dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=30, poll_interval=1, frames_per_lane=fpl,
                            msg_dtype="f16" if f16 else "f32", tune={"compact": compact, "q_order": q_order})
            for _ in range(2):                                    # the child is reused by the second call
                out, iters = dec.decode(y)
                assert np.array_equal(out, want["out"]), (compact, fpl)
                assert np.array_equal(iters, want["iters"]), (compact, fpl)
                st = dec.stats()
                assert st["frames_converged"] == n_conv, (compact, fpl)     # done flags scattered back by the child
                assert st["iterations_launched"] == 30
            dec.close()


@pytest.mark.parametrize("algo,f16", [("sp", False), ("ms", False), ("ms", True)])
def test_device_side_tail_for_asynchronous_callers(built, algo, f16):
    """Early termination WITHOUT host polling (ldpc_decode_device returns at once, all rounds are
    enqueued): once few enough frames still run, a kernel moves them into the overflow tiles and
    every later launch works on those alone -- decided on the device.  Bits, iteration counts and
    the converged count must be the oracle's; with the feature off and with a threshold that is
    reached late the same."""
    import torch
    g, og, K, M, z = _graph(codes.RATE_1_2, 1152)
    B = 2100                                                      # 9 tiles of 256 / 33 of 64
    rng = np.random.default_rng(43)
    y = channel.awgn_frames(1152, 0, B, 0.62, seed=44)            # most frames converge within a few rounds ...
    hard = rng.choice(B, 90, replace=False)
    y[hard] = channel.awgn_frames(1152, 7000, 90, 1.05, seed=45)  # ... some scattered ones (almost) never do
    want = oracle.decode(og, y, algo, max_iter=30, msg_f16=f16)
    rows, cols = codes.wimax_edges(codes.RATE_1_2, 1152)
    n_conv = int(converged_frames(rows, cols, M, want["hard"]).sum())
    assert (want["iters"] == 30).sum() >= 65 and (want["iters"] < 12).sum() > B - 120
    yd = torch.from_numpy(y).cuda()
    nb = L.out_bytes(K, B)
    for tune in ({}, {"compact": 100}, {"device_tail": False}):
        for fpl in (1, 2, 4):
            dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=30, poll_interval=0, frames_per_lane=fpl,
                            msg_dtype="f16" if f16 else "f32", tune=tune)
            for _ in range(2):
                out = torch.full((nb,), 0xEE, dtype=torch.uint8, device="cuda")
                it = torch.zeros(B, dtype=torch.int32, device="cuda")
                dec.decode_device(yd.data_ptr(), B, out.data_ptr(), nb, it.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy(), want["out"]), (tune, fpl)
                assert np.array_equal(it.cpu().numpy(), want["iters"]), (tune, fpl)
                st = dec.stats()
                assert st["frames_converged"] == n_conv and st["batch_time"] == 30, (tune, fpl)
                assert st["iterations_launched"] == 30              # asynchronous: every round is enqueued
            dec.close()
    # Idle hint: an asynchronous decoder remembers the previous call's iteration count and launches later
    # rounds with fewer, fatter workgroups.  Results must not depend on it -- neither when the hint is
    # right (the same easy batch again) nor when it is wrong (a batch with stragglers right after).
    y_easy = channel.awgn_frames(1152, 9000, B, 0.6, seed=46)
    want_easy = oracle.decode(og, y_easy, algo, max_iter=30, msg_f16=f16)
    assert want_easy["iters"].max() < 20
    for fpl in (1, 4):
        dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=30, poll_interval=0, frames_per_lane=fpl,
                        msg_dtype="f16" if f16 else "f32")
        for yy, ww in ((y_easy, want_easy), (y_easy, want_easy), (y, want), (y_easy, want_easy)):
            ydv = torch.from_numpy(yy).cuda()
            out = torch.zeros(nb, dtype=torch.uint8, device="cuda")
            it = torch.zeros(B, dtype=torch.int32, device="cuda")
            dec.decode_device(ydv.data_ptr(), B, out.data_ptr(), nb, it.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), ww["out"]) and np.array_equal(it.cpu().numpy(), ww["iters"]), fpl
        dec.close()
    # a batch too small for the overflow area simply runs without it
    dec = L.Decoder(g, K, max_batch=300, algo=algo, max_iter=30, msg_dtype="f16" if f16 else "f32")
    o2, i2 = dec.decode(y[:300])
    assert np.array_equal(i2, want["iters"][:300]) and np.array_equal(o2, want["out"][:300 * K // 8])
    dec.close()


@pytest.mark.parametrize("path", TDMPHOST, ids=lambda p: p.split("tdmphost_")[-1][:-4])
@pytest.mark.parametrize("V", [1, 2, 4])
def test_host_layered_bit_exact_vs_reference_host_path(built, path, V):
    """LDPC_ALGO_LAYERED_HOST (DecodeTDMP where the reference's host-layered path is well defined:
    seeds 2/3A and 5/6, all rows of one weight) against what the reference's own *TDMP kernels
    produced under its host loop (MyLdpc.cpp:889-976): bytes, all hard bits, `Time=`, converged
    count, and the messages and posteriors of one iteration, 0 ulp."""
    gd = load_golden(path)
    g, og, K, M, z = _graph(int(gd["rate"]), int(gd["N"]))
    y, tap = gd["y"], int(gd["tap_iter"])
    B = y.shape[0]
    dec = L.Decoder(g, K, max_batch=B, algo="layered_host", max_iter=int(gd["times"]), layer_rows=z, frames_per_lane=V)
    out, iters = dec.decode(y)
    assert np.array_equal(out, gd["out"])
    st = dec.stats()
    assert st["batch_time"] == int(gd["time"]) and st["frames_converged"] == int((gd["flags"] == 0).sum())
    o = oracle.decode(og, y, "layered_host", max_iter=int(gd["times"]), layer_rows=z)
    assert np.array_equal(iters, o["iters"])
    assert np.array_equal(dec.dump(3, B).astype(np.uint8), gd["hard"])
    dec.set_tap(tap)
    dec.decode(y)
    run = np.nonzero(o["iters"] >= tap)[0]
    assert np.array_equal(dec.dump(0, B)[run], gd["tap_r"][run], equal_nan=True)
    assert np.array_equal(dec.dump(2, B)[run], gd["tap_post"][run], equal_nan=True)
    dec.close()


def test_host_layered_degenerate_values_and_the_cpp_class(built, tmp_path):
    """Zeros, ties, infinities and NaNs through LDPC_ALGO_LAYERED_HOST against the oracle (the
    three-way hard decision keeps a bit where P is 0 or NaN), a ragged multi-group batch, and
    Coder::addDecodeType(DecodeTDMP) choosing this path for rate 5/6 and 2/3A."""
    g, og, K, M, z = _graph(codes.RATE_5_6, 960)
    rng = np.random.default_rng(23)
    y = channel.awgn_frames(960, 0, 150, 0.45, seed=23)
    y[0, rng.choice(960, 200, replace=False)] = 0.0
    y[1, :] = 0.0
    y[2, rng.choice(960, 30, replace=False)] = np.inf
    y[3, rng.choice(960, 30, replace=False)] = -np.inf
    y[4, rng.choice(960, 5, replace=False)] = np.nan
    y[5, :] = 1.0
    y[6, :] = -1.0
    y[7, rng.choice(960, 300, replace=False)] *= 400.0
    want = oracle.decode(og, y, "layered_host", max_iter=15, layer_rows=z)
    dec = L.Decoder(g, K, max_batch=64, algo="layered_host", max_iter=15, layer_rows=z)
    out, iters = dec.decode(y)
    assert np.array_equal(out, want["out"]) and np.array_equal(iters, want["iters"])
    dec.close()
    exe = str(tmp_path / "coder_roundtrip")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "coder_roundtrip.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "myldpccppapi_amd"), "-lmyldpc", "-lldpc_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "myldpccppapi_amd")])
    for rate, N, K2, snr in ((5, 960, 800, "8"), (1, 672, 448, "5.5")):
        pre = str(tmp_path / ("tdmp%d" % rate))
        r = subprocess.run([exe, str(rate), str(N), "3000", "16", snr, "TDMP", "7", "--dump", pre], capture_output=True, text=True)
        assert r.returncode == 0 and "ErrNum=0" in r.stdout, r.stdout + r.stderr
        g2, og2, Kc, Mc, zc = _graph(rate, N)
        assert Kc == K2
        post = np.fromfile(pre + ".post", np.float32).reshape(-1, N)
        ref = oracle.decode(og2, post, "layered_host", max_iter=40, layer_rows=zc)["out"]
        assert np.array_equal(np.fromfile(pre + ".out", np.uint8), ref[:3000])
