"""The CPU restatement (oracle/) against the golden vectors made from the
reference's own kernel source (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

import oracle
from util import assert_taps_equal, golden_files, load_golden, wimax_oracle_graph

FLOOD = golden_files("flood")
LAYERED = golden_files("layered")
MSCL = golden_files("mscl")
TDMPHOST = golden_files("tdmphost")


def test_fixtures_present():
    assert len(FLOOD) >= 8 and len(LAYERED) >= 7


@pytest.mark.parametrize("path", FLOOD, ids=lambda p: p.split("flood_")[-1][:-4])
@pytest.mark.parametrize("algo", ["ms", "sp"])
def test_flooding_matches_reference_kernels(path, algo):
    gd = load_golden(path)
    g, rows, cols, K, M, z = wimax_oracle_graph(int(gd["rate"]), int(gd["N"]))
    assert g.E == int(gd["E"]) and K == int(gd["K"])
    tap = int(gd["tap_iter"])
    o = oracle.decode(g, gd["y"], algo, max_iter=int(gd["times"]), tap_iter=tap)
    assert np.array_equal(o["out"], gd[algo + "_out"])          # packed bytes, bit for bit
    assert np.array_equal(o["hard"], gd[algo + "_hard"])        # all N hard bits
    assert int(o["iters"].max()) == int(gd[algo + "_time"])     # the batch's `Time=`
    # a frame's flag stays set exactly when its syndrome never became clean
    converged = (gd[algo + "_flags"] == 0)
    assert np.array_equal(o["iters"][~converged], np.full((~converged).sum(), int(gd["times"])))
    # messages of iteration `tap`, on frames that reached it
    if algo == "ms":
        assert_taps_equal(o["taps"]["r"], gd["ms_tap_r"], o["iters"] >= tap, "lR")
        assert_taps_equal(o["taps"]["post"], gd["ms_tap_post"], o["iters"] >= tap, "lPostP")
        assert_taps_equal(o["taps"]["q"], gd["ms_tap_q"], o["iters"] > tap, "lQ")
    else:
        assert_taps_equal(o["taps"]["r0"], gd["sp_tap_r0"], o["iters"] >= tap, "r0")
        assert_taps_equal(o["taps"]["r1"], gd["sp_tap_r1"], o["iters"] >= tap, "r1")
        assert_taps_equal(o["taps"]["q0"], gd["sp_tap_q0"], o["iters"] > tap, "q0")
        assert_taps_equal(o["taps"]["q1"], gd["sp_tap_q1"], o["iters"] > tap, "q1")


@pytest.mark.parametrize("path", LAYERED, ids=lambda p: p.split("layered_")[-1][:-4])
def test_layered_matches_fused_reference_kernel(path):
    gd = load_golden(path)
    g, rows, cols, K, M, z = wimax_oracle_graph(int(gd["rate"]), int(gd["N"]))
    o = oracle.decode(g, gd["y"], "layered", max_iter=int(gd["times"]), layer_rows=z)
    # frames on which the reference reads an uninitialised variable are outside the contract
    assert not o["undefined"].any()
    assert np.array_equal(o["out"], gd["out"])


@pytest.mark.parametrize("path", MSCL, ids=lambda p: p.split("mscl_")[-1][:-4])
def test_fused_flooding_matches_reference_kernel(path):
    """decodeOnceMS (DecodeMSCL): product sign, 1000/1001 two-minimum rule, 120 iterations."""
    gd = load_golden(path)
    g, rows, cols, K, M, z = wimax_oracle_graph(int(gd["rate"]), int(gd["N"]))
    o = oracle.decode(g, gd["y"], "ms_fused", max_iter=int(gd["times"]))
    assert not o["undefined"].any()
    assert np.array_equal(o["out"], gd["out"])


@pytest.mark.parametrize("path", TDMPHOST, ids=lambda p: p.split("tdmphost_")[-1][:-4])
def test_host_layered_matches_reference_host_path(path):
    """DecodeTDMP as the reference's HOST drives it (MyLdpc.cpp:889-976 over the *TDMP kernels,
    decodeCL.c:203-300), on the seeds whose rows all have one weight (2/3A, 5/6) -- the only ones
    for which the reference's layer sizes are right: bytes, all hard bits, `Time=`, flags and one
    iteration's messages and posteriors, bit for bit."""
    gd = load_golden(path)
    g, rows, cols, K, M, z = wimax_oracle_graph(int(gd["rate"]), int(gd["N"]))
    tap = int(gd["tap_iter"])
    o = oracle.decode(g, gd["y"], "layered_host", max_iter=int(gd["times"]), layer_rows=z, tap_iter=tap)
    assert np.array_equal(o["out"], gd["out"])
    assert np.array_equal(o["hard"], gd["hard"])
    assert int(o["iters"].max()) == int(gd["time"])
    converged = gd["flags"] == 0
    assert (o["iters"][~converged] == int(gd["times"])).all()
    assert_taps_equal(o["taps"]["r"], gd["tap_r"], o["iters"] >= tap, "lR")
    assert_taps_equal(o["taps"]["post"], gd["tap_post"], o["iters"] >= tap, "lPostP")


def test_host_layered_is_refused_where_the_reference_is_wrong():
    g, *_ = wimax_oracle_graph(0, 576)          # rate 1/2: row weights 6 and 7
    with pytest.raises(ValueError):
        oracle.decode(g, np.ones((1, 576), np.float32), "layered_host", layer_rows=24)


def test_ms_equals_cpu_decoder_packing_when_k_is_byte_aligned():
    """decodeCPU (pack_mode 1) and the MS kernel chain (pack_mode 0) give the same bytes
    when K % 8 == 0."""
    gd = load_golden([p for p in FLOOD if "c576_34b" in p][0])
    g, *_ = wimax_oracle_graph(int(gd["rate"]), int(gd["N"]))
    a = oracle.decode(g, gd["y"], "ms", pack_mode=0)
    b = oracle.decode(g, gd["y"], "ms", pack_mode=1)
    assert np.array_equal(a["out"], b["out"])


def test_graph_facts_from_the_survey_probe():
    """E of the reference's own H builder for four (rate, N) pairs."""
    gd = load_golden(golden_files("graph")[0])
    for rate, N, E in gd["cases"]:
        rows, cols = oracle.wimax_edges(int(rate), int(N))
        assert len(rows) == int(E)
