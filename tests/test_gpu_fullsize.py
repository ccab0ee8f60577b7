"""BASELINE.json's full-size code on the GPU: DVB-S2-profile (64800, 32400).
Bit-exact against the oracle on sampled frames, plus size-independent properties
(codeword symmetry, noiseless idempotence, all-zero decode)."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle
import myldpccppapi_amd as L
from myldpccppapi_amd import channel, codes

pytestmark = pytest.mark.gpu
N, K = 64800, 32400
M = N - K


@pytest.fixture(scope="module")
def code():
    rows, cols = codes.dvbs2_profile_edges(N, K)
    return rows, cols, L.Graph(rows, cols, M, N), oracle.Graph(rows, cols, M, N, K)


def _ira_encode(rows, cols, info_bits):
    """Systematic IRA encoding of the profile code: p_m = p_{m-1} ^ (A s)_m."""
    sel = cols < K
    acc = np.zeros(M, np.int64)
    np.add.at(acc, rows[sel], info_bits[cols[sel]].astype(np.int64))
    p = np.cumsum(acc & 1) & 1
    cw = np.concatenate([info_bits, p.astype(np.uint8)])
    syn = np.zeros(M, np.int64)
    np.add.at(syn, rows, cw[cols].astype(np.int64))
    assert not (syn & 1).any()
    return cw


def test_sampled_frames_bit_exact_sp_50_iterations(built, code):
    """configs[1] shape: SP fp32, 50 iterations.  260 frames = two ragged V=4 tiles;
    a converging and a non-converging operating point; 3 frames of each are checked
    against the CPU oracle bit for bit (the oracle needs ~1 s per frame here)."""
    rows, cols, g, og = code
    for sigma, seed in ((0.55, 31), (0.95, 32)):
        y = channel.awgn_frames(N, 0, 260, sigma, seed=seed)
        dec = L.Decoder(g, K, max_batch=260, algo="sp", max_iter=50)
        out, iters = dec.decode(y)
        pick = [0, 131, 259]
        o = oracle.decode(og, y[pick], "sp", max_iter=50)
        kb = K // 8
        for i, f in enumerate(pick):
            assert np.array_equal(out[f * kb:(f + 1) * kb], o["out"][i * kb:(i + 1) * kb]), (sigma, f)
            assert iters[f] == o["iters"][i]
        if sigma < 0.6:
            # the all-zero codeword is recovered wherever the syndrome became clean; a few
            # frames saturate to 0/0 = NaN in the fp32 probability domain and never converge
            # (the reference's behaviour, SURVEY.md "hard part 2") -- the oracle agrees on them
            conv = iters < 50
            assert conv.mean() > 0.9
            assert not out.reshape(260, kb)[conv].any()
        else:
            assert iters.min() == 50
        dec.close()


def test_codeword_symmetry_full_size(built, code):
    """Min-sum is sign-symmetric in exact fp32: decoding (noise on codeword c) equals
    (decoding the same noise on the all-zero word) XOR c -- a check at full size that
    needs no oracle run."""
    algo = "ms"
    rows, cols, g, og = code
    rng = np.random.default_rng(5)
    B = 96
    y0 = channel.awgn_frames(N, 0, B, 0.62, seed=41)
    cw = np.stack([_ira_encode(rows, cols, rng.integers(0, 2, K).astype(np.uint8)) for _ in range(B)])
    yc = (y0 * channel.bpsk(cw)).astype(np.float32)
    dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=30, layer_rows=360)
    out0, it0 = dec.decode(y0)
    outc, itc = dec.decode(yc)
    bits0 = channel.unpack_bits(out0, K, B)
    bitsc = channel.unpack_bits(outc, K, B)
    assert np.array_equal(bitsc, bits0 ^ cw[:, :K])
    assert np.array_equal(it0, itc)
    dec.close()


def test_noiseless_codewords_are_fixed_points(built, code):
    rows, cols, g, og = code
    rng = np.random.default_rng(6)
    cw = np.stack([_ira_encode(rows, cols, rng.integers(0, 2, K).astype(np.uint8)) for _ in range(8)])
    y = channel.bpsk(cw)
    for algo in ("sp", "ms"):     # (consecutive rows of the IRA staircase share columns: no layering)
        dec = L.Decoder(g, K, max_batch=8, algo=algo, max_iter=50)
        out, iters = dec.decode(y)
        assert np.array_equal(channel.unpack_bits(out, K, 8), cw[:, :K]) and (iters == 1).all()
        dec.close()


def test_high_rate_check_degree_30(built):
    """configs[4] shape: (64800, 58320), check degree 30 -> generic check kernel; min-sum
    against the oracle on two frames."""
    N2, K2 = 64800, 58320
    rows, cols = codes.dvbs2_profile_edges(N2, K2)
    g = L.Graph(rows, cols, N2 - K2, N2)
    og = oracle.Graph(rows, cols, N2 - K2, N2, K2)
    y = channel.awgn_frames(N2, 0, 66, 0.36, seed=51)
    kb = K2 // 8
    for f16 in (False, True):        # configs[4] stores messages in fp16
        dec = L.Decoder(g, K2, max_batch=66, algo="ms", max_iter=30, msg_dtype="f16" if f16 else "f32")
        out, iters = dec.decode(y)
        o = oracle.decode(og, y[[0, 65]], "ms", max_iter=30, msg_f16=f16)
        assert np.array_equal(out[:kb], o["out"][:kb]) and np.array_equal(out[65 * kb:], o["out"][kb:])
        assert iters[0] == o["iters"][0] and iters[65] == o["iters"][1]
        dec.close()


def test_layered_bg1_profile_z384(built):
    """configs[3] shape: BG1-profile QC code, Z = 384 (N = 26112, E = 121344), layered
    min-sum with 384-row layers.  Two frames against the oracle bit for bit, and the
    codeword symmetry over the whole batch."""
    Z = 384
    base = codes.nr_bg1_profile_base(Z=Z)
    rows, cols = codes.qc_edges(base, Z)
    Nb, Kb, Mb = 68 * Z, 22 * Z, 46 * Z
    g = L.Graph(rows, cols, Mb, Nb)
    og = oracle.Graph(rows, cols, Mb, Nb, Kb)
    rng = np.random.default_rng(7)
    B = 70
    y0 = channel.awgn_frames(Nb, 0, B, 0.8, seed=61)
    cw = np.stack([codes.nr_bg1_profile_encode(base, Z, rng.integers(0, 2, Kb).astype(np.uint8))
                   for _ in range(B)])
    syn = np.zeros(Mb, np.int64)
    np.add.at(syn, rows, cw[0][cols].astype(np.int64))
    assert not (syn & 1).any()
    yc = (y0 * channel.bpsk(cw)).astype(np.float32)
    dec = L.Decoder(g, Kb, max_batch=B, algo="layered", max_iter=20, layer_rows=Z)
    out0, it0 = dec.decode(y0)
    outc, itc = dec.decode(yc)
    assert np.array_equal(channel.unpack_bits(outc, Kb, B), channel.unpack_bits(out0, Kb, B) ^ cw[:, :Kb])
    assert np.array_equal(it0, itc)
    o = oracle.decode(og, y0[[0, 69]], "layered", max_iter=20, layer_rows=Z)
    kb = Kb // 8
    assert np.array_equal(out0[:kb], o["out"][:kb]) and np.array_equal(out0[69 * kb:], o["out"][kb:])
    assert it0[0] == o["iters"][0] and it0[69] == o["iters"][1]
    dec.set_tap(2)
    dec.decode(y0)
    r_ldsp, p_ldsp = dec.dump(0, B), dec.dump(2, B)
    dec.close()
    # the default above is layered_ldsp_kernel (posterior in LDS); the one-launch-per-layer
    # streaming kernels must give the same bits, iteration counts and messages for all 70 frames
    dec = L.Decoder(g, Kb, max_batch=B, algo="layered", max_iter=20, layer_rows=Z, tune={"ldsp": False})
    out1, it1 = dec.decode(y0)
    assert np.array_equal(out1, out0) and np.array_equal(it1, it0)
    dec.set_tap(2)
    dec.decode(y0)
    run = np.nonzero(it0 >= 2)[0]
    assert np.array_equal(dec.dump(0, B)[run], r_ldsp[run]) and np.array_equal(dec.dump(2, B)[run], p_ldsp[run])
    dec.close()
    # fewer persistent workgroups than frames
    dec = L.Decoder(g, Kb, max_batch=B, algo="layered", max_iter=20, layer_rows=Z, tune={"ldsp": True, "ldsp_grid": 16})
    out2, it2 = dec.decode(y0)
    assert np.array_equal(out2, out0) and np.array_equal(it2, it0)
    dec.close()
    # all 68 block columns in LDS (104 KB of dynamic LDS, one workgroup per CU), alive together with a
    # decoder of a small code that uses the same kernel with 10 KB
    big = L.Decoder(g, Kb, max_batch=B, algo="layered", max_iter=20, layer_rows=Z, tune={"ldsp": True, "ldsp_ext": False})
    zs = 96
    rs, cs = codes.qc_edges(codes.nr_bg1_profile_base(Z=zs), zs)
    gs = L.Graph(rs, cs, 46 * zs, 68 * zs)
    small = L.Decoder(gs, 22 * zs, max_batch=9, algo="layered", max_iter=20, layer_rows=zs)
    ys = channel.awgn_frames(68 * zs, 0, 9, 0.8, seed=62)
    outs, its = small.decode(ys)
    out3, it3 = big.decode(y0)
    assert np.array_equal(out3, out0) and np.array_equal(it3, it0)
    os_ = oracle.decode(oracle.Graph(rs, cs, 46 * zs, 68 * zs, 22 * zs), ys, "layered", max_iter=20, layer_rows=zs)
    assert np.array_equal(outs, os_["out"]) and np.array_equal(its, os_["iters"])
    big.close()
    small.close()


def test_full_batch_4096_frames_indexing(built, code):
    """configs[1] at its real size: 4096 frames = 16 tiles of 256, message arrays of 3.7 GB each
    (element offsets beyond 2^31).  Sum-product, 12 iterations; frames from the first, a middle
    and the last tile are checked bit for bit against the oracle, and every frame must report
    the iteration count the oracle's sample shows for non-converging noise."""
    rows, cols, g, og = code
    B = 4096
    rng = np.random.default_rng(77)
    y = (1.0 + 0.95 * rng.standard_normal((B, N), dtype=np.float32)).astype(np.float32)
    dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=12)
    out, iters = dec.decode(y)
    pick = [0, 255, 2048 + 77, 4095 - 256, 4095]
    o = oracle.decode(og, y[pick], "sp", max_iter=12)
    kb = K // 8
    for i, f in enumerate(pick):
        assert np.array_equal(out[f * kb:(f + 1) * kb], o["out"][i * kb:(i + 1) * kb]), f
        assert iters[f] == o["iters"][i]
    assert (iters == 12).all() and dec.stats()["frames_converged"] == 0
    dec.close()
    del y


def test_tail_compaction_at_full_size(built, code):
    """configs[1] dimensions, min-sum with early termination and host polling: 4096 frames of which
    all but 37 converge within a few rounds; the stragglers (spread over all 16 tiles, element
    offsets beyond 2^31 in the gather) are finished by the child decoder.  Every frame's iteration
    count against the oracle's for a sample, the stragglers' bits and counts against the oracle."""
    rows, cols, g, og = code
    B = 4096
    rng = np.random.default_rng(78)
    y = (1.0 + 0.72 * rng.standard_normal((B, N), dtype=np.float32)).astype(np.float32)
    slow = np.sort(rng.choice(B, 37, replace=False))
    y[slow] = (1.0 + 0.86 * rng.standard_normal((37, N), dtype=np.float32)).astype(np.float32)
    dec = L.Decoder(g, K, max_batch=B, algo="ms", max_iter=25, poll_interval=1)
    out, iters = dec.decode(y)
    st = dec.stats()
    pick = np.r_[slow[[0, 5, 18, 36]], [0, 255, 2048, 4095]]
    o = oracle.decode(og, y[pick], "ms", max_iter=25)
    kb = K // 8
    for i, f in enumerate(pick):
        assert np.array_equal(out[f * kb:(f + 1) * kb], o["out"][i * kb:(i + 1) * kb]), f
        assert iters[f] == o["iters"][i], f
    fast = np.setdiff1d(np.arange(B), slow)
    assert iters[fast].max() < 25 and iters[slow].min() > iters[fast].max()        # the premise of the test
    assert st["frames_converged"] == int((iters < 25).sum())
    assert st["iterations_launched"] == int(iters.max())
    dec.close()


def test_bg1_layered_at_its_full_batch_of_8192(built):
    """BASELINE.json configs[3] at its real size: BG1-profile Z = 384 (N = 26112, E = 121344), layered
    min-sum, batch 8192, device buffers.  The persistent grid of the record kernel (768 workgroups)
    wraps over the frames ~11 times.  Checked: the codeword symmetry decode(noise on c) =
    decode(noise on 0) XOR c with equal iteration counts on every converging frame of the batch (16
    distinct codewords, 8192 distinct noise frames, 7680 of them at a converging noise level), and
    frames 0, 767, 768, 4100, 7700 and 8191 bit for bit against the oracle."""
    import torch
    Z = 384
    base = codes.nr_bg1_profile_base(Z=Z)
    rows, cols = codes.qc_edges(base, Z)
    Nb, Kb, Mb = 68 * Z, 22 * Z, 46 * Z
    g = L.Graph(rows, cols, Mb, Nb)
    og = oracle.Graph(rows, cols, Mb, Nb, Kb)
    B = 8192
    rng = np.random.default_rng(17)
    cws = np.stack([codes.nr_bg1_profile_encode(base, Z, rng.integers(0, 2, Kb).astype(np.uint8)) for _ in range(16)])
    which = rng.integers(0, 16, B)
    bits = torch.from_numpy(cws[which]).cuda()                               # [B, N] code bits
    # 7680 frames that converge within a few iterations, then 512 that never do
    y0 = torch.empty((B, Nb), dtype=torch.float32, device="cuda")            # noise on the all-zero word
    for lo, hi, sd in ((0, 7680, 0.9), (7680, B, 1.2)):
        channel.awgn_device(Nb, lo, hi - lo, sd, seed=63, out=y0[lo:hi])
    yc = y0 * (1.0 - 2.0 * bits.to(torch.float32))                           # the mirrored noise on the codewords
    torch.cuda.synchronize()
    dec = L.Decoder(g, Kb, max_batch=B, algo="layered", max_iter=20, layer_rows=Z)
    nb = L.out_bytes(Kb, B)
    outs, its = [], []
    for y in (y0, yc):
        out = torch.empty(nb, dtype=torch.uint8, device="cuda")
        it = torch.empty(B, dtype=torch.int32, device="cuda")
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), nb, it.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
        its.append(it.cpu().numpy())
    st = dec.stats()
    b0 = channel.unpack_bits(outs[0], Kb, B)
    bc = channel.unpack_bits(outs[1], Kb, B)
    # exact sign symmetry needs a decode without exact ties (bit = P < 0 maps +0 and -0 to the same
    # bit): it holds on every frame that converges; the frames that never do are covered by the oracle
    conv = its[0] < 20
    assert np.array_equal(bc[conv], (b0 ^ cws[which][:, :Kb])[conv])
    # (an exact tie P = 0 in an intermediate iteration can move the clean syndrome by one round)
    assert (its[0][conv] != its[1][conv]).mean() < 0.005
    assert (its[1][~conv] == 20).mean() > 0.99
    assert (its[0][:7680] < 20).mean() > 0.99 and (its[0][7680:] == 20).mean() > 0.99     # both kinds of frames
    assert st["frames"] == B and st["batch_time"] == 20
    pick = [0, 767, 768, 4100, 7700, 8191]
    o = oracle.decode(og, y0[pick].cpu().numpy(), "layered", max_iter=20, layer_rows=Z)
    kb = Kb // 8
    for i, f in enumerate(pick):
        assert o["undefined"][i] == 0
        assert np.array_equal(outs[0][f * kb:(f + 1) * kb], o["out"][i * kb:(i + 1) * kb]), f
        assert its[0][f] == o["iters"][i], f
    dec.close()


def test_rate_910_fp16_early_termination_at_its_full_batch_of_4096(built):
    """BASELINE.json configs[4] at its real size: DVB-S2-profile (64800, 58320), fp16 message storage,
    early termination with host polling and tail compaction -- and without polling, with the device-side
    hand-over -- batch 4096.  Iteration counts of ALL frames and all bytes equal an unpolled,
    uncompacted run; 264 frames (the first tile + 8 scattered
    ones, the last included) equal the oracle's fp16-message min-sum (frames decoded in threads)."""
    N2, K2 = 64800, 58320
    rows, cols = codes.dvbs2_profile_edges(N2, K2)
    g = L.Graph(rows, cols, N2 - K2, N2)
    og = oracle.Graph(rows, cols, N2 - K2, N2, K2)
    B = 4096
    y = channel.awgn_device(N2, 0, B, 0.34, seed=64).cpu().numpy()
    hard_idx = [5, 1300, 2222, 4095]
    y[hard_idx] = channel.awgn_device(N2, 9000, 4, 0.46, seed=65).cpu().numpy()      # stragglers -> the child
    ref = L.Decoder(g, K2, max_batch=B, algo="ms", max_iter=50, msg_dtype="f16", poll_interval=0,
                    tune={"compact": -1, "device_tail": False})
    out_ref, it_ref = ref.decode(y)
    st_ref = ref.stats()
    ref.close()
    dec = L.Decoder(g, K2, max_batch=B, algo="ms", max_iter=50, msg_dtype="f16", poll_interval=2)
    out, iters = dec.decode(y)
    st = dec.stats()
    dec.close()
    assert np.array_equal(iters, it_ref) and np.array_equal(out, out_ref)
    assert iters[hard_idx].min() > np.delete(iters, hard_idx).max()                    # the premise: they ran on alone
    # the asynchronous form: no polling, hand-over decided on the device (overflow tiles), then again
    # with the idle hint of the first call
    adec = L.Decoder(g, K2, max_batch=B, algo="ms", max_iter=50, msg_dtype="f16", poll_interval=0)
    for _ in range(2):
        out_a, it_a = adec.decode(y)
        assert np.array_equal(it_a, it_ref) and np.array_equal(out_a, out_ref)
        assert adec.stats()["frames_converged"] == st_ref["frames_converged"]
    adec.close()
    assert st["frames_converged"] == st_ref["frames_converged"] >= int((iters < 50).sum())
    pick = np.r_[np.arange(256), [1300, 2047, 2048, 2222, 3000, 3839, 3840, 4095]]
    kb = K2 // 8
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        res = list(ex.map(lambda f: oracle.decode(og, y[f:f + 1], "ms", max_iter=50, msg_f16=True), pick))
    for f, o in zip(pick, res):
        assert np.array_equal(out[f * kb:(f + 1) * kb], o["out"]), f
        assert iters[f] == o["iters"][0], f


def test_hand_over_of_hundreds_of_frames_is_bit_exact(built, code):
    """Round 3's large hand-over: once a poll has seen a tenth of the frames finished the host polls every round, and up to
    1024 running frames (a quarter of a 4096-frame batch) move to the child through the row-wise LDS gather and the
    atomics-free way back (compact_gather_rows_kernel, compact_hard_back_kernel).  (a) configs[4]'s own operating point --
    rate 9/10, fp16, sigma 0.43: 681 frames still run after round 5 -- and (b) the headline code, fp32 min-sum, 4096 frames
    of which 700 are noisier.  Bytes and iteration counts of ALL frames equal an unpolled, uncompacted run; fewer
    tile-rounds do work; samples equal the oracle."""
    rows, cols, g, og = code
    cases = []
    N2, K2 = 64800, 58320
    r2, c2 = codes.dvbs2_profile_edges(N2, K2)
    cases.append(("f16", L.Graph(r2, c2, N2 - K2, N2), oracle.Graph(r2, c2, N2 - K2, N2, K2), N2, K2,
                  channel.awgn_device(N2, 0, 4096, 0.43, seed=20260101).cpu().numpy(), 50, [0, 2047, 4095]))
    rng = np.random.default_rng(91)
    y = (1.0 + 0.70 * rng.standard_normal((4096, N), dtype=np.float32)).astype(np.float32)
    slow = np.sort(rng.choice(4096, 700, replace=False))
    y[slow] = (1.0 + 0.80 * rng.standard_normal((700, N), dtype=np.float32)).astype(np.float32)
    cases.append(("f32", g, og, N, K, y, 30, [int(slow[0]), int(slow[350]), 1]))
    for msg, gg, ogg, n_, k_, yy, max_iter, pick in cases:
        ref = L.Decoder(gg, k_, max_batch=4096, algo="ms", max_iter=max_iter, msg_dtype=msg, poll_interval=0,
                        tune={"compact": -1, "device_tail": False})
        out_ref, it_ref = ref.decode(yy)
        fr_ref = ref.stats()["frame_rounds"]
        ref.close()
        dec = L.Decoder(gg, k_, max_batch=4096, algo="ms", max_iter=max_iter, msg_dtype=msg, poll_interval=2)
        out, iters = dec.decode(yy)
        st = dec.stats()
        dec.close()
        assert np.array_equal(iters, it_ref) and np.array_equal(out, out_ref), msg
        # the premise: hundreds of frames were still running when at most a quarter was left
        hist = np.bincount(iters, minlength=max_iter + 1)
        left_after = 4096 - np.cumsum(hist)                   # frames running after round r
        assert ((left_after >= 128) & (left_after <= 1024)).any(), (msg, hist)
        assert st["frame_rounds"] < 0.95 * fr_ref, (msg, st["frame_rounds"], fr_ref)
        kb = k_ // 8
        for f in pick:
            o = oracle.decode(ogg, yy[f:f + 1], "ms", max_iter=max_iter, msg_f16=(msg == "f16"))
            assert np.array_equal(out[f * kb:(f + 1) * kb], o["out"]) and iters[f] == o["iters"][0], (msg, f)


def test_sum_product_hand_over_of_hundreds_of_frames_carries_the_hard_bits(built):
    """The sum-product rule keeps a column's previous bit on a tie or a NaN (decodeCL.c:78-82), so a hand-over to the child
    must carry the hard bits (min-sum decides anew every round and does not): a (3600, 1800) IRA code with the headline
    code's kernel set, 4096 frames of which 650 are noisier -- some of them flood with NaN --, polled every round, child
    with tiles of 256 frames.  All bytes and iteration counts equal an uncompacted run; 96 frames equal the oracle."""
    Ns, Ks = 3600, 1800
    rows, cols = codes.dvbs2_profile_edges(Ns, Ks, profile=[(8, 2), (3, 3)])
    g = L.Graph(rows, cols, Ns - Ks, Ns)
    og = oracle.Graph(rows, cols, Ns - Ks, Ns, Ks)
    rng = np.random.default_rng(92)
    y = channel.awgn_frames(Ns, 0, 4096, 0.50, seed=92)
    slow = np.sort(rng.choice(4096, 650, replace=False))
    y[slow] = channel.awgn_frames(Ns, 5000, 650, 0.66, seed=93)
    ref = L.Decoder(g, Ks, max_batch=4096, algo="sp", max_iter=30, poll_interval=0, tune={"compact": -1, "device_tail": False})
    out_ref, it_ref = ref.decode(y)
    fr_ref = ref.stats()["frame_rounds"]
    ref.close()
    dec = L.Decoder(g, Ks, max_batch=4096, algo="sp", max_iter=30, poll_interval=1)
    out, iters = dec.decode(y)
    st = dec.stats()
    dec.close()
    assert np.array_equal(iters, it_ref) and np.array_equal(out, out_ref)
    left_after = 4096 - np.cumsum(np.bincount(iters, minlength=31))
    assert ((left_after >= 128) & (left_after <= 1024)).any(), left_after
    assert st["frame_rounds"] < fr_ref
    pick = np.r_[slow[:48], np.setdiff1d(np.arange(4096), slow)[:48]]
    o = oracle.decode(og, y[pick], "sp", max_iter=30)
    kb = Ks // 8
    for i, f in enumerate(pick):
        assert np.array_equal(out[f * kb:(f + 1) * kb], o["out"][i * kb:(i + 1) * kb]) and iters[f] == o["iters"][i], f


def test_cpp_coder_at_the_reference_constructible_full_size(built, tmp_path):
    """The largest code the reference's own constructor can make at this length:
    Coder(32400, 64800, rate_1_2) -- z = 2700, E = 205200 (tests/golden/graph_facts.npz) -- through
    the C++ class end to end like Test.cpp: 256 frames, structured encoder at N = 64800, Coder::test
    channel, SP / MS / CPU decode.  ParityFail = 0, ErrNum = 0, and the MS and CPU bytes equal the
    oracle's decode of the very same postCode floats (32 frames)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "coder_roundtrip")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "coder_roundtrip.cpp"), "-o", exe,
                           "-L" + os.path.join(root, "myldpccppapi_amd"), "-lmyldpc", "-lldpc_hip",
                           "-Wl,-rpath," + os.path.join(root, "myldpccppapi_amd")])
    frames, kb = 256, 32400 // 8
    src_len = frames * kb
    rows, cols = codes.wimax_edges(codes.RATE_1_2, 64800)
    assert len(rows) == 205200
    og = oracle.Graph(rows, cols, 32400, 64800, 32400)
    for mode, snr in (("SP", "6"), ("MS", "3"), ("CPU", "3")):
        pre = str(tmp_path / mode)
        out = subprocess.run([exe, "0", "64800", str(src_len), "256", snr, mode, "3", "--dump", pre],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        fields = dict(l.split("=", 1) for l in out.stdout.split() if "=" in l)
        assert fields["NonZeros"] == "205200" and fields["ParityFail"] == "0" and fields["ErrNum"] == "0", out.stdout
        if mode == "SP":
            continue
        post = np.fromfile(pre + ".post", np.float32).reshape(frames, 64800)
        got = np.fromfile(pre + ".out", np.uint8)
        with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
            res = list(ex.map(lambda f: oracle.decode(og, post[f:f + 1], "ms", max_iter=40)["out"], range(32)))
        assert np.array_equal(got[:32 * kb], np.concatenate(res)), mode
        assert int(fields["Time"]) < 40


def test_cpp_coder_round_trip_with_frames_not_byte_aligned(built, tmp_path):
    """Coder(324, 648, rate_1_2): K % 8 = 4.  encode() reads frame f at source byte (f*K)/8 and
    decode() writes it there (MyLdpc.cpp:556-564, decodeCL.c:191-192), so over 5 frames the source
    bytes 80 and 161 belong to no frame: a noiseless-ish round trip returns the payload except for
    exactly those bytes (zero) -- the reference's behaviour, kept."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "coder_roundtrip")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "coder_roundtrip.cpp"), "-o", exe,
                           "-L" + os.path.join(root, "myldpccppapi_amd"), "-lmyldpc", "-lldpc_hip",
                           "-Wl,-rpath," + os.path.join(root, "myldpccppapi_amd")])
    src_len = 167
    src = np.array([ord("a") + i % 26 for i in range(src_len)], np.uint8)
    for mode in ("SP", "MS", "TDMPCL"):
        for batch in ("8", "2"):
            pre = str(tmp_path / ("k4_%s_%s" % (mode, batch)))
            out = subprocess.run([exe, "0", "648", str(src_len), batch, "9", mode, "4", "--dump", pre],
                                 capture_output=True, text=True)
            assert out.returncode == 0 and "ParityFail=0" in out.stdout, out.stdout + out.stderr
            got = np.fromfile(pre + ".out", np.uint8)
            bad = np.nonzero(got != src)[0]
            # frames end at a 4-bit boundary: the last half byte of each frame's K bits is not packed (toChar
            # writes K/8 whole bytes), so the only differences are the two orphan bytes
            assert bad.tolist() == [80, 161], (mode, batch, bad)
            assert got[80] == 0 and got[161] == 0


def test_hbm_probe_reports_both_cache_policies(built):
    """ldpc_hbm_probe_device: a 256 MiB float4 copy with the default cache policy and with non-temporal
    accesses; the headline figure is the better of the two and both are plausible HBM rates."""
    best, plain, nt = L.capi.hbm_probe(0, 256 << 20, 3, by_policy=True)
    assert best == max(plain, nt)
    assert 1000.0 < plain < 8000.0 and 1000.0 < nt < 8000.0, (plain, nt)
    with pytest.raises(L.LdpcError):
        L.capi.hbm_probe(0, 1 << 10, 3)                                     # below 1 MiB: refused
    # the same copy back to back for 50 ms: a plausible rate too, not above the burst figure by more than noise
    sustained = L.capi.hbm_sustained(0, 256 << 20, 50)
    assert 1000.0 < sustained < 1.15 * best, (sustained, best)


def test_placement_search_keeps_results_and_reports_what_it_saw(built, code):
    """cfg.tune_place: a decoder with large arrays tries fresh allocations for its two message arrays when it is created
    and keeps the fastest combination.  Results must not depend on it; the report must be sane; and
    `place = 1` must skip the search."""
    rows, cols, g, og = code
    y = channel.awgn_frames(N, 0, 300, 0.9, seed=77)
    ref = None
    for place in (1, 0, 3):
        dec = L.Decoder(g, K, max_batch=1024, algo="ms", max_iter=12, tune={"place": place})
        rep = dec.placement()
        if place == 1:
            assert rep is None
        else:
            assert rep is not None and 1 <= len(rep["candidates_ms"]) <= 15 and 0 <= rep["kept"] < len(rep["candidates_ms"])
            assert all(0.01 < t < 50 for t in rep["candidates_ms"])
            assert rep["candidates_ms"][rep["kept"]] == min(rep["candidates_ms"])
        out, iters = dec.decode(y)
        if ref is None:
            ref = (out, iters)
            o = oracle.decode(og, y[:2], "ms", max_iter=12)
            assert np.array_equal(out[:2 * K // 8], o["out"]) and np.array_equal(iters[:2], o["iters"])
        assert np.array_equal(out, ref[0]) and np.array_equal(iters, ref[1]), place
        assert set(dec.array_addresses()) == {"Q", "R", "chan", "hard"}
        dec.close()


def test_single_group_host_call_is_cut_in_two_and_gives_the_same_bytes(built, code):
    """ldpc_decode cuts a large call that is ONE launch group (frames <= max_batch, >= 2048 frames, >= 256 MiB of channel
    values, K a multiple of 8) into two groups so that the second half's copy runs beside the first half's decode.
    Bytes and iteration counts must equal the device-buffer path, which decodes the batch as one group; 2100 frames:
    groups of 1280 and 820, a ragged second one.  Both input modes; then the same handle with more frames than
    max_batch (full-size groups again: the slots must still hold them)."""
    import torch
    rows, cols, g, og = code
    frames = 2100
    yd = channel.awgn_device(N, 0, frames, 0.66, seed=77)
    nb = L.out_bytes(K, frames)
    ref = L.Decoder(g, K, max_batch=4096, algo="ms", max_iter=25, poll_interval=2)
    out_d = torch.empty(nb, dtype=torch.uint8, device="cuda")
    it_d = torch.empty(frames, dtype=torch.int32, device="cuda")
    ref.decode_device(yd.data_ptr(), frames, out_d.data_ptr(), nb, it_d.data_ptr(), None)
    torch.cuda.synchronize()
    want_out, want_it = out_d.cpu().numpy(), it_d.cpu().numpy()
    want_st = ref.stats()
    ref.close()
    assert (want_it < 25).sum() > frames // 2            # a decoding operating point, not noise
    yh = yd.cpu().numpy()
    for mode in ("staged", "lock_pages"):
        dec = L.Decoder(g, K, max_batch=4096, algo="ms", max_iter=25, poll_interval=2, host_input=mode)
        out, iters = dec.decode(yh)
        assert np.array_equal(out, want_out) and np.array_equal(iters, want_it), mode
        st = dec.stats()                                 # the call's counts cover both groups
        for k in ("frames", "frames_converged", "batch_time", "iterations_launched"):
            assert st[k] == want_st[k], (mode, k, st[k], want_st[k])
        assert 0 < st["frame_rounds"] <= want_st["frame_rounds"] * 1.2
        dec.close()
    small = L.Decoder(g, K, max_batch=2048, algo="ms", max_iter=25, poll_interval=2)
    out, iters = small.decode(yh[:2048])                 # one group of exactly max_batch: cut into 1024 + 1024
    kb = K // 8
    assert np.array_equal(out, want_out[:2048 * kb]) and np.array_equal(iters, want_it[:2048])
    out, iters = small.decode(yh)                        # 2100 > max_batch: groups of 2048 + 52
    assert np.array_equal(out, want_out) and np.array_equal(iters, want_it)
    small.close()
    assert L.capi.host_locked_ranges() == (0, 0)
