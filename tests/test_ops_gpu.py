"""Per-op parity on the GPU: every call goes through the C-ABI (librn_hip.so) and is
checked against the CPU oracle on the same seeded inputs.

Tolerances.  Integer/index results (argmax, shapes) and pure data movement / max /
relu / add are bit-exact.  Contractions on the matrix-core path sum the same
products in a different (fixed) k order than the reference's sequential chain, so
they are compared with rtol 2e-5 / atol 2e-5 * sqrt(K)-scaled magnitude; the direct
fallback kernel keeps the reference order and is bit-exact with the oracle.
Batch-norm follows the reference's double-precision expression: <= 1 ulp.
"""
import ctypes

import numpy as np
import pytest

import resnet_c_amd as R
from oracle import oracle as O
from resnet_c_amd import ops

pytestmark = pytest.mark.gpu


def rnd(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def assert_close(got, want, k_terms):
    scale = float(np.abs(want).max()) + 1e-6
    tol = 3e-7 * np.sqrt(k_terms) * scale + 1e-6
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape
    assert err <= tol, f"max err {err:.3e} > tol {tol:.3e} (K={k_terms})"


def test_reference_test_cu_patterns(golden_dir):
    import os
    k = np.load(os.path.join(golden_dir, "ops_kat.npz"))
    # small-integer data: every path must be exact
    assert np.array_equal(ops.conv2d(k["conv_x"], k["conv_w"]), k["conv_y"])
    assert np.array_equal(ops.linear(k["lin_x"], k["lin_w"], k["lin_b"]), k["lin_y"])
    assert np.array_equal(ops.relu(k["relu_x"]), k["relu_y"])


# B, Cin, Cout, H, W, k, stride, pad  -- direct kernel (Cin % 32 != 0, not stem-shaped)
DIRECT = [(2, 5, 4, 8, 8, 3, 1, 1), (1, 8, 8, 5, 5, 1, 1, 0), (3, 8, 4, 6, 6, 1, 2, 0),
          (2, 6, 6, 7, 9, 3, 2, 1), (1, 33, 3, 4, 4, 2, 1, 0), (1, 5, 2, 3, 3, 3, 1, 2)]


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("case", DIRECT)
def test_conv_direct_is_bit_exact_with_reference_order(case, layout):
    B, Cin, Cout, H, W, k, s, p = case
    x, w = rnd((B, Cin, H, W), 1 + sum(case)), rnd((Cout, Cin, k, k), 2 + sum(case))
    assert np.array_equal(ops.conv2d(x, w, s, p, layout), O.conv2d(x, w, s, p))


# matrix-core path: Cin % 32 == 0, and the small-Cin stem form (Cin <= 4, k <= 8)
GEMM = [
    (2, 64, 64, 12, 12, 1, 1, 0),    # 1x1, one K tile pair, BN=64
    (1, 64, 256, 14, 14, 1, 1, 0),   # 1x1 wide output
    (2, 256, 64, 9, 9, 1, 1, 0),     # ragged M (162 rows): row guards
    (1, 128, 128, 10, 10, 3, 1, 1),  # 3x3 pad 1
    (2, 64, 64, 13, 11, 3, 2, 1),    # 3x3 stride 2, odd sizes
    (1, 256, 512, 8, 8, 1, 2, 0),    # projection shortcut: 1x1 stride 2
    (2, 32, 96, 7, 7, 3, 1, 1),      # Cout not a multiple of the tile (column guards)
    (1, 32, 40, 5, 6, 5, 1, 2),      # 5x5
    (2, 3, 64, 32, 32, 7, 2, 3),     # stem form
    (1, 3, 64, 224, 224, 7, 2, 3),   # the real stem shape, 98 M tiles
    (1, 4, 16, 9, 9, 3, 1, 1),       # stem form with Cin = 4
    (1, 1, 8, 10, 10, 8, 1, 3),      # stem form, k = 8
    (1, 2, 3, 4, 4, 2, 1, 0),        # tiny
]


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("case", GEMM)
def test_conv_gemm_matches_oracle(case, layout):
    B, Cin, Cout, H, W, k, s, p = case
    if layout == "nhwc" and Cin < 4:
        pytest.skip("NHWC small-Cin input takes the direct kernel (covered above)")
    x, w = rnd((B, Cin, H, W), 11 + sum(case)), rnd((Cout, Cin, k, k), 12 + sum(case))
    assert_close(ops.conv2d(x, w, s, p, layout), O.conv2d(x, w, s, p), Cin * k * k)


def test_conv_zero_padding_is_not_read_as_garbage():
    # an all-ones image: border outputs count exactly the in-bounds taps
    x = np.ones((1, 32, 6, 6), dtype=np.float32)
    w = np.ones((32, 32, 3, 3), dtype=np.float32)
    got = ops.conv2d(x, w, 1, 1, "nhwc")
    assert got[0, 0, 0, 0] == 32 * 4 and got[0, 0, 0, 3] == 32 * 6 and got[0, 0, 3, 3] == 32 * 9
    xs = np.ones((1, 3, 10, 10), dtype=np.float32)
    ws = np.ones((64, 3, 7, 7), dtype=np.float32)
    gs = ops.conv2d(xs, ws, 2, 3)
    assert gs[0, 0, 0, 0] == 3 * 16 and gs[0, 5, 2, 2] == 3 * 49


@pytest.mark.parametrize("case", [(2, 64, 64, 12, 12, 3, 1, 1), (1, 128, 256, 7, 7, 1, 1, 0),
                                  (2, 3, 64, 20, 20, 7, 2, 3), (1, 6, 5, 6, 6, 3, 1, 1)])
def test_fused_epilogue_matches_unfused_sequence(case):
    B, Cin, Cout, H, W, k, s, p = case
    seed = 100 + sum(case)
    x, w = rnd((B, Cin, H, W), seed), rnd((Cout, Cin, k, k), seed + 1)
    g = np.random.default_rng(seed + 2)
    gamma, beta = g.random(Cout, dtype=np.float32) + 0.5, g.standard_normal(Cout, dtype=np.float32)
    mean, var = g.standard_normal(Cout, dtype=np.float32), g.random(Cout, dtype=np.float32) + 0.5
    y = O.conv2d(x, w, s, p)
    res = rnd(y.shape, seed + 3)
    want = O.relu_(O.add_(O.batchnorm2d(y, gamma, beta, mean, var), res))
    sc = (gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + 1e-5))
    scale = sc.astype(np.float32)
    shift = (beta.astype(np.float64) - mean.astype(np.float64) * sc).astype(np.float32)
    got = ops.conv2d_nhwc_fused(x, w, s, p, scale, shift, res, True)
    assert_close(got, want, Cin * k * k + 4)


@pytest.mark.parametrize("case", [
    # B, Cin, Cout, H, W, k, pad, Cin2, stride2   (first conv stride 1, as conv3 of a block)
    (2, 64, 256, 14, 14, 1, 0, 64, 1),     # layer1.0: conv3 + downsample, both 1x1 stride 1
    (2, 128, 512, 7, 7, 1, 0, 256, 2),     # layer2.0: downsample reads every second pixel
    (1, 32, 40, 5, 6, 3, 1, 96, 2),        # 3x3 first conv, ragged M and Cout, odd second image
    (3, 64, 64, 9, 9, 1, 0, 32, 1),
])
def test_conv_pair_is_the_sum_of_both_branches(case):
    """rn_conv2d_nhwc_pair_forward_dt: relu(bn3(conv3(t)) + bnd(convd(x))) from one K loop.  The
    scales are folded into the weight rows, so the reference applies them the same way."""
    B, Cin, Cout, H, W, k, p, Cin2, s2 = case
    seed = 300 + sum(case)
    t, w = rnd((B, Cin, H, W), seed), rnd((Cout, Cin, k, k), seed + 1)
    H2, W2 = (H - 1) * s2 + 1 + (s2 - 1), (W - 1) * s2 + 1   # ragged: last row unused when s2 = 2
    x2, w2 = rnd((B, Cin2, H2, W2), seed + 2), rnd((Cout, Cin2, 1, 1), seed + 3)
    g = np.random.default_rng(seed + 4)
    sc1, sc2 = g.random(Cout, dtype=np.float32) + 0.5, g.random(Cout, dtype=np.float32) + 0.5
    shift = g.standard_normal(Cout, dtype=np.float32)
    a = O.conv2d(t, w * sc1[:, None, None, None], 1, p)
    b = O.conv2d(x2, w2 * sc2[:, None, None, None], s2, 0)
    assert a.shape == b.shape
    want = O.relu_(a + b + shift[None, :, None, None])
    got = ops.conv2d_nhwc_pair(t, w, x2, w2, 1, p, s2, sc1, sc2, shift, None, True)
    assert_close(got, want, Cin * k * k + Cin2 + 4)
    # without scales / shift / relu, plus a residual
    res = rnd(a.shape, seed + 5)
    want2 = O.conv2d(t, w, 1, p) + O.conv2d(x2, w2, s2, 0) + res
    got2 = ops.conv2d_nhwc_pair(t, w, x2, w2, 1, p, s2, residual=res)
    assert_close(got2, want2, Cin * k * k + Cin2 + 4)


def test_conv_pair_rejects_what_it_cannot_do():
    t, w = rnd((1, 32, 4, 4), 1), rnd((32, 32, 1, 1), 2)
    with pytest.raises(R.RnError):   # second channel count not a multiple of 32
        ops.conv2d_nhwc_pair(t, w, rnd((1, 16, 4, 4), 3), rnd((32, 16, 1, 1), 4))
    with pytest.raises(R.RnError):   # second convolution gives another output size
        ops.conv2d_nhwc_pair(t, w, rnd((1, 32, 9, 9), 3), rnd((32, 32, 1, 1), 4), stride2=3)


@pytest.mark.parametrize("case", [
    (2, 3, 64, 224, 224, 7, 2, 3),   # the stem: K = 147 -> 160
    (3, 3, 64, 20, 23, 7, 2, 3),     # ragged M, odd width
    (1, 1, 8, 9, 9, 3, 1, 1),        # one channel, K = 9 -> 32 (one K tile)
    (2, 4, 72, 12, 10, 5, 1, 2),     # K = 100 -> 128, Cout not a tile multiple
    (2, 2, 16, 8, 8, 2, 2, 0),       # no padding, K = 8
])
def test_exact_small_cin_form(case):
    """rn_conv2d_nhwc_exact_forward: K = k*k*Cin packed without slot padding, dword gathers
    from a physically padded image."""
    B, Cin, Cout, H, W, k, s, p = case
    seed = 500 + sum(case)
    x, w = rnd((B, Cin, H, W), seed), rnd((Cout, Cin, k, k), seed + 1)
    want = O.conv2d(x, w, s, p)
    assert_close(ops.conv2d_nhwc_exact(x, w, s, p), want, Cin * k * k)
    g = np.random.default_rng(seed + 2)
    sc, sh = g.random(Cout, dtype=np.float32) + 0.5, g.standard_normal(Cout, dtype=np.float32)
    got = ops.conv2d_nhwc_exact(x, w, s, p, sc, sh, None, True)
    assert_close(got, O.relu_(want * sc[None, :, None, None] + sh[None, :, None, None]),
                 Cin * k * k + 4)


def test_batchnorm_fold_entry_point():
    import ctypes
    from resnet_c_amd import _lib as L
    C = 70
    g = np.random.default_rng(5)
    w, b = g.random(C, dtype=np.float32) + 0.5, g.standard_normal(C, dtype=np.float32)
    m, v = g.standard_normal(C, dtype=np.float32), g.random(C, dtype=np.float32) + 0.5
    ctx = R.get_ctx()
    T = [R.FloatTensor.from_numpy(a, R.Device.GPU) for a in (w, b, m, v)]
    sc, sh = R.FloatTensor((C,), R.Device.GPU), R.FloatTensor((C,), R.Device.GPU)
    L.check(L.lib().rn_batchnorm2d_fold(ctx.handle, *(t.data() for t in T), sc.data(), sh.data(), C),
            "fold", ctx.handle)
    s64 = w.astype(np.float64) / np.sqrt(v.astype(np.float64) + 1e-5)
    assert np.array_equal(sc.numpy(), s64.astype(np.float32))
    np.testing.assert_allclose(sh.numpy(), (b - m * s64).astype(np.float32), rtol=0, atol=1e-7)


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("case", [(2, 64, 112, 112, 3, 2, 1), (2, 6, 9, 9, 3, 2, 1), (1, 8, 7, 7, 2, 2, 0),
                                  (3, 5, 8, 6, 3, 1, 1), (2, 2048, 7, 7, 7, 1, 0), (1, 12, 7, 7, 7, 1, 0),
                                  (3, 5, 13, 24, 3, 2, 1), (1, 300, 7, 7, 7, 1, 0), (2, 3, 6, 8, 3, 2, 1)])
def test_pools_bit_exact(case, layout):
    B, C, H, W, k, s, p = case
    x = rnd((B, C, H, W), 7 + sum(case))
    assert np.array_equal(ops.maxpool2d(x, k, s, p, layout), O.maxpool2d(x, k, s, p))
    # same tap order and the same two divisions as the reference: bit-exact
    assert np.array_equal(ops.avgpool2d(x, k, s, p, layout), O.avgpool2d(x, k, s, p))


def test_maxpool_all_padding_window_and_nan():
    x = np.full((1, 4, 3, 3), -5.0, dtype=np.float32)
    x[0, 0, 1, 1] = np.nan  # fmax suppresses NaN like the reference's device fmax
    for layout in ("nchw", "nhwc"):
        got = ops.maxpool2d(x, 3, 2, 1, layout)
        assert np.array_equal(got, O.maxpool2d(x, 3, 2, 1)) and not np.isnan(got).any()
    # the vectorised NCHW form (width a multiple of 8): NaNs and -inf in every tap position of a lane's quad
    y = rnd((2, 3, 10, 16), 99)
    y[0, 0, 3, :] = np.nan
    y[1, 2, :, 7] = np.nan
    y[0, 1, 0:3, 0:3] = -np.inf
    for layout in ("nchw", "nhwc"):
        got = ops.maxpool2d(y, 3, 2, 1, layout)
        assert np.array_equal(got, O.maxpool2d(y, 3, 2, 1)) and not np.isnan(got).any()
    # the column walk of the NHWC form (eight or more output rows, channels a multiple of 4): odd and even
    # heights, a lane's carried row maximum all NaN / all -inf, signed zeros, several row segments
    for shape in ((2, 8, 33, 20), (1, 4, 64, 9), (3, 12, 16, 17)):
        z = rnd(shape, 101 + sum(shape))
        z[0, 0, 5, :] = np.nan
        z[0, 1, 6:9, :] = np.nan
        z[-1, 2, :, 3] = np.nan
        z[0, 3, 0:8, 0:8] = -np.inf
        z[-1, 0, 10:14, :] = 0.0
        z[-1, 0, 11, ::2] = -0.0
        got = ops.maxpool2d(z, 3, 2, 1, "nhwc")
        want = O.maxpool2d(z, 3, 2, 1)
        assert np.array_equal(got, want) and not np.isnan(got).any()
        assert np.array_equal(np.signbit(got), np.signbit(want))


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("shape", [(2, 64, 56, 56), (3, 7, 5, 5), (1, 256, 14, 14), (2, 5, 4, 6), (4, 8, 1, 1),
                                   (3, 2048, 7, 7), (70, 5, 9, 9),
                                   # NCHW, small planes and a batch to walk (one image position per lane):
                                   (16, 512, 7, 7), (9, 256, 14, 14), (8, 3, 16, 16), (12, 7, 2, 2), (64, 33, 3, 4)])
def test_batchnorm_matches_double_expression(shape, layout):
    """ops.cu:150 type by type (fp32 subtraction, then double, one rounding): the same bits as the oracle."""
    g = np.random.default_rng(sum(shape))
    x = g.standard_normal(shape, dtype=np.float32) * 3
    C = shape[1]
    w, b = g.random(C, dtype=np.float32) + 0.5, g.standard_normal(C, dtype=np.float32)
    m, v = g.standard_normal(C, dtype=np.float32), g.random(C, dtype=np.float32) + 0.5
    want = O.batchnorm2d(x, w, b, m, v)
    got = ops.batchnorm2d(x, w, b, m, v, layout, inplace=True)  # in place, like main.cu:138
    assert np.array_equal(got, want)
    assert np.array_equal(ops.batchnorm2d(x, w, b, m, v, layout, inplace=False), got)


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_batchnorm_special_values_follow_the_reference_expression(layout):
    """Infinite / NaN / huge inputs, a zero and a negative denominator: what `(x - m) / sqrt(v + 1e-5) * w + b`
    gives in the reference's types (the Markstein quotient of the kernel must not turn inf into NaN)."""
    B, C, H, W = 8, 8, 4, 4
    g = np.random.default_rng(77)
    x = g.standard_normal((B, C, H, W), dtype=np.float32)
    x[0, :, 0, 0] = np.inf
    x[1, :, 0, 1] = -np.inf
    x[2, :, 1, 0] = np.nan
    x[3, :, 1, 1] = 3.0e38
    x[4, :, 2, 2] = -3.0e38
    x[5, :, 3, 3] = 1e-42      # denormal
    w, b = g.random(C, dtype=np.float32) + 0.5, g.standard_normal(C, dtype=np.float32)
    m, v = g.standard_normal(C, dtype=np.float32), g.random(C, dtype=np.float32) + 0.5
    w[1] = 0.0                 # inf * 0 = NaN
    w[2] = -1.5
    v[3] = -1.0                # sqrt of a negative: NaN everywhere in the channel
    v[4] = np.float32(-1e-5)   # var + 1e-5 in double: the float is not exactly -1e-5, tiny or negative
    m[5] = np.inf              # x - inf in fp32
    with np.errstate(all="ignore"):
        want = O.batchnorm2d(x, w, b, m, v)
        got = ops.batchnorm2d(x, w, b, m, v, layout, inplace=False)
    assert np.array_equal(got, want, equal_nan=True)
    ok = ~np.isnan(want)
    assert np.array_equal(np.signbit(got[ok]), np.signbit(want[ok]))
    assert np.isinf(want).any() and np.isnan(want).any()


@pytest.mark.parametrize("n", [1, 3, 4, 17, 1023, 1024, 1025, 4099, 1 << 20])
def test_relu_add_any_length_in_place(n):
    a, b = rnd((n,), n), rnd((n,), n + 1)
    a[::7] = -a[::7]
    want_relu, want_add = O.relu(a), O.add(a, b)   # the oracle's restatement of ops.cu:130-137,153-160
    for inplace in (True, False):
        assert np.array_equal(ops.relu(a, inplace), want_relu)
        assert np.array_equal(ops.add(a, b, inplace), want_add)
    z = np.array([np.nan, -0.0, -1.0, 2.0], dtype=np.float32)
    assert ops.relu(z).tolist() == [0.0, 0.0, 0.0, 2.0]  # fmax(NaN, 0) == 0 (ops.cu:136)
    assert np.array_equal(ops.relu(z), O.relu(z))


@pytest.mark.parametrize("case", [(3, 16, 8), (256, 2048, 1000), (5, 64, 10), (1, 2048, 1000), (2, 7, 3), (70, 96, 130)])
def test_linear(case):
    B, fin, fout = case
    x, w, b = rnd((B, fin), sum(case)), rnd((fout, fin), 1 + sum(case)), rnd((fout,), 2)
    want = O.linear(x, w, b)
    got = ops.linear(x, w, b)
    if fin % 32:
        assert np.array_equal(got, want)  # direct kernel: reference order
    else:
        assert_close(got, want, fin)
    assert_close(ops.linear(x, w, None), O.linear(x, w, None), fin)  # bias is nullable


def test_argmax_first_maximum_and_nan_rules():
    g = np.random.default_rng(0)
    logits = g.standard_normal((9, 1000), dtype=np.float32)
    logits[1, [10, 500]] = 50.0      # tie -> lower index
    logits[2, 999] = 60.0            # last lane
    logits[3, 0] = np.nan            # NaN at 0 is never displaced
    logits[4, 77] = np.nan           # NaN elsewhere never wins
    logits[5, :] = -np.inf           # all equal -> 0
    logits[6, 64] = 70.0             # second pass of lane 0
    assert np.array_equal(ops.argmax(logits), O.argmax(logits))
    assert np.array_equal(ops.argmax(logits), R.model.argmax(logits))
    small = g.standard_normal((3, 5), dtype=np.float32)
    assert np.array_equal(ops.argmax(small), O.argmax(small))


def test_layout_converters_round_trip():
    import ctypes
    from resnet_c_amd import _lib as L
    ctx, lib = R.get_ctx(), L.lib()
    for shape in [(2, 3, 5, 7), (1, 64, 9, 9), (3, 33, 4, 4), (2, 256, 7, 7), (1, 1, 1, 1)]:
        B, C, H, W = shape
        x = rnd(shape, sum(shape))
        src = R.FloatTensor.from_numpy(x, R.Device.GPU)
        mid, back = R.FloatTensor(shape, R.Device.GPU), R.FloatTensor(shape, R.Device.GPU)
        L.check(lib.rn_nchw_to_nhwc(ctx.handle, src.data(), mid.data(), B, C, H, W), "t", ctx.handle)
        L.check(lib.rn_nhwc_to_nchw(ctx.handle, mid.data(), back.data(), B, C, H, W), "t", ctx.handle)
        ctx.sync()
        assert np.array_equal(mid.cpu()._storage.reshape(B, H, W, C), x.transpose(0, 2, 3, 1))
        assert np.array_equal(back.numpy(), x)
        for cpad in (4, 8):
            if cpad < C:
                continue
            pad = R.FloatTensor((B, H, W, cpad), R.Device.GPU)
            L.check(lib.rn_nchw_to_nhwc_pad(ctx.handle, src.data(), pad.data(), B, C, H, W, cpad), "p", ctx.handle)
            ctx.sync()
            got = pad.cpu()._storage.reshape(B, H, W, cpad)
            assert np.array_equal(got[..., :C], x.transpose(0, 2, 3, 1)) and not got[..., C:].any()


@pytest.mark.parametrize("case", [(2, 3, 224, 224, 3, 3), (3, 3, 10, 13, 3, 3), (2, 3, 6, 6, 4, 1),
                                  (1, 1, 5, 5, 1, 2), (2, 2, 8, 8, 4, 0), (1, 3, 7, 9, 8, 2)])
def test_bordered_nhwc_image_fp32(case):
    """rn_nchw_to_nhwc_pad_dt(F32): [B,H+2b,W+2b,Cpad] with zero border and zero pad channels
    (16-byte-store kernel where the image size allows, element kernel otherwise)."""
    from resnet_c_amd import _lib as L
    B, C, H, W, cpad, border = case
    ctx, lib = R.get_ctx(), L.lib()
    x = rnd((B, C, H, W), 900 + sum(case))
    src = R.FloatTensor.from_numpy(x, R.Device.GPU)
    Hp, Wp = H + 2 * border, W + 2 * border
    dst = R.FloatTensor((B, Hp, Wp, cpad), R.Device.GPU)
    L.check(lib.rn_memset(ctx.handle, dst.data(), 0xFF, B * Hp * Wp * cpad * 4), "memset", ctx.handle)
    L.check(lib.rn_nchw_to_nhwc_pad_dt(ctx.handle, L.RN_DTYPE_F32, src.data(), dst.data(), B, C, H, W,
                                       cpad, border), "pad_dt", ctx.handle)
    ctx.sync()
    want = np.zeros((B, Hp, Wp, cpad), dtype=np.float32)
    want[:, border:border + H, border:border + W, :C] = x.transpose(0, 2, 3, 1)
    assert np.array_equal(dst.cpu()._storage.reshape(B, Hp, Wp, cpad), want)


@pytest.mark.parametrize("case", [
    (1, 512, 512, 7, 7, 3, 1, 1),      # layer4 conv2 at B = 1: 8 tiles, 144 K tiles
    (2, 1024, 256, 14, 14, 1, 1, 0),   # layer3 conv1
    (1, 256, 72, 5, 5, 3, 2, 1),       # ragged M and Cout, stride 2
    (3, 2048, 1000, 1, 1, 1, 1, 0),    # the fc shape
])
def test_split_k_latency_mode(case):
    """rn_ctx_set_split_k: under-filled contractions split their K loop over several blocks;
    the partial sums are added in split order by a second kernel that also runs the epilogue.
    Same tolerance as the unsplit kernel, deterministic, and off again afterwards."""
    B, Cin, Cout, H, W, k, s, p = case
    seed = 700 + sum(case)
    x, w = rnd((B, Cin, H, W), seed), rnd((Cout, Cin, k, k), seed + 1)
    g = np.random.default_rng(seed + 2)
    sc, sh = g.random(Cout, dtype=np.float32) + 0.5, g.standard_normal(Cout, dtype=np.float32)
    y = O.conv2d(x, w, s, p)
    res = rnd(y.shape, seed + 3)
    want = O.relu_(y * sc[None, :, None, None] + sh[None, :, None, None] + res)
    ctx = R.get_ctx()
    plain = ops.conv2d_nhwc_fused(x, w, s, p, sc, sh, res, True)
    ctx.set_split_k(16)
    try:
        got = ops.conv2d_nhwc_fused(x, w, s, p, sc, sh, res, True)
        again = ops.conv2d_nhwc_fused(x, w, s, p, sc, sh, res, True)
        raw = ops.conv2d_nhwc_fused(x, w, s, p)   # no epilogue at all
    finally:
        ctx.set_split_k(0)
    assert_close(got, want, Cin * k * k + 4)
    assert_close(raw, y, Cin * k * k)
    assert np.array_equal(got, again)
    assert np.array_equal(ops.conv2d_nhwc_fused(x, w, s, p, sc, sh, res, True), plain)


def test_chunked_k_sum_is_the_same_whole_or_in_pieces():
    """fp32 layers with K >= 1024 add their products as ((c0 + c1) + c2) + ... over eight K chunks.
    A launch whose last round would leave most CUs idle cuts its tail tiles into (tile, chunk)
    pieces and adds them in a second kernel; tiles computed whole fold the chunks in registers.
    Same bits either way, whatever the batch size or the tile shape."""
    from resnet_c_amd import _lib as L
    Cin, Cout, H, W = 1024, 256, 14, 14
    w = rnd((Cout, Cin, 1, 1), 811)
    g = np.random.default_rng(812)
    sc, sh = g.random(Cout, dtype=np.float32) + 0.5, g.standard_normal(Cout, dtype=np.float32)
    big = rnd((36, Cin, H, W), 813)          # 444 tiles of 64x64: 188 tail tiles are cut
    got = ops.conv2d_nhwc_fused(big, w, 1, 0, sc, sh, None, True)
    want = O.relu_(O.conv2d(big[30:], w, 1, 0) * sc[None, :, None, None] + sh[None, :, None, None])
    assert_close(got[30:], want, Cin + 4)
    small = ops.conv2d_nhwc_fused(big[34:36], w, 1, 0, sc, sh, None, True)   # 28 tiles, all whole
    assert np.array_equal(got[34:36], small)
    assert np.array_equal(got[:2], ops.conv2d_nhwc_fused(big[:2], w, 1, 0, sc, sh, None, True))
    ctx, lib = R.get_ctx(), L.lib()
    try:
        for cand in range(1, lib.rn_conv_tile_candidates() + 1):
            lib.rn_ctx_set_conv_tile(ctx.handle, cand)
            assert np.array_equal(ops.conv2d_nhwc_fused(big, w, 1, 0, sc, sh, None, True), got), cand
    finally:
        lib.rn_ctx_set_conv_tile(ctx.handle, 0)


@pytest.mark.parametrize("case", [(35, 1024, 256, 14, 14, 1, 0),    # ragged last M tile inside the cut tail
                                  (35, 1024, 264, 14, 14, 1, 0),    # ragged N tile, tail not a whole tile row
                                  (18, 128, 128, 28, 28, 3, 1)])    # 3x3, K = 1152: 442 tiles, 186 cut
def test_cut_tails_match_oracle_and_small_batches(case):
    B, Cin, Cout, H, W, k, p = case
    seed = 830 + sum(case)
    x, w = rnd((B, Cin, H, W), seed), rnd((Cout, Cin, k, k), seed + 1)
    g = np.random.default_rng(seed + 2)
    sc, sh = g.random(Cout, dtype=np.float32) + 0.5, g.standard_normal(Cout, dtype=np.float32)
    res = rnd((B, Cout, H, W), seed + 3)
    got = ops.conv2d_nhwc_fused(x, w, 1, p, sc, sh, res, True)
    tailrows = slice(B - 2, B)
    want = O.relu_(O.conv2d(x[tailrows], w, 1, p) * sc[None, :, None, None]
                   + sh[None, :, None, None] + res[tailrows])
    assert_close(got[tailrows], want, Cin * k * k + 4)
    # the last images sit in the cut tail of the big launch and in whole tiles of a small one
    small = ops.conv2d_nhwc_fused(x[tailrows], w, 1, p, sc, sh, res[tailrows], True)
    assert np.array_equal(got[tailrows], small)
    assert np.array_equal(got[:1], ops.conv2d_nhwc_fused(x[:1], w, 1, p, sc, sh, res[:1], True))


@pytest.mark.parametrize("seed", range(6))
def test_chunked_layers_random_geometry(seed):
    """Random batch / image / channel sizes around the point where a launch starts cutting its
    tail (256-700 tiles): every launch geometry must give the oracle's numbers, and the first
    and last image must not depend on the others."""
    g = np.random.default_rng(9000 + seed)
    k = int(g.choice([1, 3]))
    Cin = int(g.choice([1024, 1536])) if k == 1 else int(g.choice([128, 160]))   # K >= 1024
    Cout = int(g.choice([64, 72, 128, 200, 256]))
    H, W = int(g.integers(5, 15)), int(g.integers(5, 15))
    tiles_per_image = (H * W / 64.0) * np.ceil(Cout / 64.0)
    B = int(np.clip(g.integers(256, 700) / tiles_per_image, 2, 64))
    x, w = rnd((B, Cin, H, W), 9100 + seed), rnd((Cout, Cin, k, k), 9200 + seed)
    sc, sh = g.random(Cout, dtype=np.float32) + 0.5, g.standard_normal(Cout, dtype=np.float32)
    got = ops.conv2d_nhwc_fused(x, w, 1, k // 2, sc, sh, None, True)
    for rows in (slice(0, 1), slice(B - 1, B)):
        want = O.relu_(O.conv2d(x[rows], w, 1, k // 2) * sc[None, :, None, None] + sh[None, :, None, None])
        assert_close(got[rows], want, Cin * k * k + 4)
        assert np.array_equal(got[rows], ops.conv2d_nhwc_fused(x[rows], w, 1, k // 2, sc, sh, None, True))


def test_empty_inputs_are_no_ops():
    """Zero-sized work (B = 0, N = 0, no channels out) returns RN_OK without touching the
    pointers, as a launch with an empty grid would in the reference."""
    from resnet_c_amd import _lib as L
    ctx, lib = R.get_ctx(), L.lib()
    h = ctx.handle
    assert lib.rn_conv2d_forward(h, None, None, None, 3, 1, 1, 8, 8, 0, 16, 16, 8, 8) == L.RN_OK
    assert lib.rn_conv2d_nhwc_forward(h, None, None, None, 3, 1, 1, 8, 8, 0, 32, 32, 8, 8, None) == L.RN_OK
    assert lib.rn_conv2d_nhwc_forward_dt(h, L.RN_DTYPE_BF16, L.RN_DTYPE_BF16, None, None, None, 1, 1, 0,
                                         4, 4, 0, 64, 64, 4, 4, None) == L.RN_OK
    assert lib.rn_conv2d_nhwc_exact_forward(h, None, None, None, 7, 2, 112, 112, 0, 3, 64, 230, 230,
                                            None) == L.RN_OK
    assert lib.rn_maxpool2d_forward(h, None, None, 3, 2, 1, 4, 4, 0, 8, 8, 8) == L.RN_OK
    assert lib.rn_avgpool2d_forward(h, None, None, 7, 1, 0, 1, 1, 2, 0, 7, 7) == L.RN_OK
    assert lib.rn_linear_forward(h, None, None, None, None, 0, 2048, 1000) == L.RN_OK
    assert lib.rn_batchnorm2d_forward(h, None, None, None, None, None, None, 0, 8, 16) == L.RN_OK
    assert lib.rn_add_forward(h, None, None, None, 0) == L.RN_OK
    assert lib.rn_relu_forward(h, None, None, 0) == L.RN_OK
    assert lib.rn_nchw_to_nhwc(h, None, None, 0, 3, 4, 4) == L.RN_OK
    assert lib.rn_nchw_to_nhwc_pad_dt(h, L.RN_DTYPE_F32, None, None, 0, 3, 4, 4, 3, 3) == L.RN_OK
    assert lib.rn_conv2d_pack_weight(h, None, None, 32, 0, 3) == L.RN_OK


def test_oversized_tensors_are_refused_before_any_launch():
    """Index arithmetic in the kernels is 32-bit (Shape::numel() in the reference overflows
    silently at 2^31, tensor.cuh:28): sizes past that come back as RN_ERR_INVALID.  The
    pointers are never dereferenced, so small buffers stand in for the huge tensors."""
    from resnet_c_amd import _lib as L
    ctx, lib = R.get_ctx(), L.lib()
    h = ctx.handle
    a, b, c = (R.FloatTensor((64,), R.Device.GPU) for _ in range(3))
    big = 1 << 16   # 65536 x 65536 image = 2^32 elements
    assert lib.rn_conv2d_forward(h, a.data(), b.data(), c.data(), 1, 1, 0, big, big, 1, 1, 1, big,
                                 big) == L.RN_ERR_INVALID
    assert lib.rn_conv2d_nhwc_forward(h, a.data(), b.data(), c.data(), 1, 1, 0, big, big, 1, 32, 32,
                                      big, big, None) == L.RN_ERR_INVALID
    assert lib.rn_linear_forward(h, a.data(), b.data(), c.data(), None, 1 << 20, 1 << 12,
                                 1 << 12) == L.RN_ERR_INVALID
    assert lib.rn_conv2d_nhwc_exact_forward(h, a.data(), b.data(), c.data(), 7, 2, (big - 7) // 2 + 1,
                                            (big - 7) // 2 + 1, 1, 3, 64, big, big,
                                            None) == L.RN_ERR_INVALID
    assert b"2^31" in lib.rn_last_error(h) or b"too large" in lib.rn_last_error(h)
    # kernel_size 0 / stride 0 are precondition failures, not divisions by zero
    assert lib.rn_conv2d_forward(h, a.data(), b.data(), c.data(), 0, 1, 0, 4, 4, 1, 4, 4, 4,
                                 4) == L.RN_ERR_INVALID
    assert lib.rn_conv2d_forward(h, a.data(), b.data(), c.data(), 1, 0, 0, 4, 4, 1, 4, 4, 4,
                                 4) == L.RN_ERR_INVALID


def test_error_convention_status_not_abort():
    from resnet_c_amd import _lib as L
    ctx, lib = R.get_ctx(), L.lib()
    t = R.FloatTensor((16,), R.Device.GPU)
    # in-place conv is a precondition failure: status + message, process survives
    st = lib.rn_conv2d_forward(ctx.handle, t.data(), t.data(), t.data(), 1, 1, 0, 1, 1, 1, 4, 4, 1, 1)
    assert st == L.RN_ERR_INVALID and b"in place" in lib.rn_last_error(ctx.handle)
    assert lib.rn_relu_forward(ctx.handle, None, None, 0) == L.RN_OK  # empty tensor is a no-op
    assert lib.rn_ctx_set_layout(ctx.handle, 7) == L.RN_ERR_INVALID
    with pytest.raises(R.RnError):
        R.FloatTensor.loadToCuda("/nonexistent/weights_bin/conv1.weight")
    # reference behaviour on request: sync + error check after every op
    ctx.set_sync_each_op(True)
    try:
        assert np.array_equal(ops.relu(np.array([-1, 2], dtype=np.float32)), [0, 2])
    finally:
        ctx.set_sync_each_op(False)


def test_device_tensor_file_round_trip(tmp_path):
    from resnet_c_amd import _lib as L
    x = rnd((3, 5), 9)
    p = tmp_path / "layer1.0.conv1.weight"
    x.tofile(p)
    t = R.FloatTensor.loadToCuda(str(p))
    assert t.shape() == R.Shape((15,)) and t.device == R.Device.GPU
    assert np.array_equal(t.view((3, 5)).numpy(), x)
    ctx = R.get_ctx()
    out = tmp_path / "dump.bin"
    L.check(L.lib().rn_save_f32_file(ctx.handle, str(out).encode(), t.data(), 15), "save", ctx.handle)
    assert np.array_equal(np.fromfile(out, dtype=np.float32), x.reshape(-1))
    with pytest.raises(RuntimeError):
        t.toDevice(R.Device.GPU)  # same-device copy is unsupported (tensor.cuh:193)


def test_packed_weight_cache_of_the_nchw_route():
    """rn_ctx_set_weight_cache: rn_conv2d_forward (the reference's OIHW / NCHW signature) packs
    a weight buffer once instead of per call.  Same bits with the cache on or off; an entry dies
    when the buffer is written through rn_memcpy_h2d or freed."""
    from resnet_c_amd import _lib as L
    ctx, lib = R.get_ctx(), L.lib()
    x = rnd((2, 64, 9, 9), 901)
    w1, w2 = rnd((96, 64, 3, 3), 902), rnd((96, 64, 3, 3), 903)
    xd = R.FloatTensor.from_numpy(x, R.Device.GPU)
    wd = R.FloatTensor.from_numpy(w1, R.Device.GPU)
    out = R.FloatTensor((2, 96, 9, 9), R.Device.GPU)

    def run():
        L.check(lib.rn_conv2d_forward(ctx.handle, xd.data(), out.data(), wd.data(), 3, 1, 1, 9, 9, 2, 64, 96, 9, 9),
                "conv", ctx.handle)
        ctx.sync()
        return out.numpy()

    plain1 = run()
    assert_close(plain1, O.conv2d(x, w1, 1, 1), 64 * 9)
    ctx.set_weight_cache(True)
    try:
        assert np.array_equal(run(), plain1)      # packs and remembers
        assert np.array_equal(run(), plain1)      # reuses the panel
        # new values in the same buffer: the entry must die with the write
        L.check(lib.rn_memcpy_h2d(ctx.handle, wd.data(), w2.ctypes.data, w2.nbytes), "h2d", ctx.handle)
        got2 = run()
        assert_close(got2, O.conv2d(x, w2, 1, 1), 64 * 9)
        assert not np.array_equal(got2, plain1)
        # a write that starts INSIDE the cached buffer (its second half) stales the panel as well
        half = w2.size // 2
        w3 = w2.copy().reshape(-1)
        w3[half:] = w1.reshape(-1)[half:]
        L.check(lib.rn_memcpy_h2d(ctx.handle, wd.data() + 4 * half, w3[half:].ctypes.data, 4 * (w3.size - half)),
                "h2d", ctx.handle)
        got3 = run()
        assert_close(got3, O.conv2d(x, w3.reshape(w2.shape), 1, 1), 64 * 9)
        assert not np.array_equal(got3, got2)
        del wd                                     # rn_free of a cached weight buffer
        wd = R.FloatTensor.from_numpy(w1, R.Device.GPU)
        assert np.array_equal(run(), plain1)
    finally:
        ctx.set_weight_cache(False)
    assert np.array_equal(run(), plain1)


def test_one_thread_two_contexts_interleaved():
    """A host thread may own several contexts (on a multi-GPU node: on several devices); every
    entry point binds its context's device first.  Interleaved calls on two contexts give what
    each gives alone."""
    a, b = R.Context(0), R.Context(0)
    try:
        from resnet_c_amd import _lib as L
        lib = L.lib()
        x = rnd((1 << 16,), 77)
        bufs = []
        for c in (a, b):
            p = ctypes.c_void_p()
            L.check(lib.rn_malloc(c.handle, ctypes.byref(p), x.nbytes), "malloc", c.handle)
            L.check(lib.rn_memcpy_h2d(c.handle, p, x.ctypes.data, x.nbytes), "h2d", c.handle)
            bufs.append(p)
        for _ in range(3):
            for c, p in zip((a, b), bufs):
                L.check(lib.rn_relu_forward(c.handle, p, p, x.size), "relu", c.handle)
                L.check(lib.rn_add_forward(c.handle, p, p, p, x.size), "add", c.handle)
        want = O.relu(x)
        for _ in range(3):
            want = O.add(O.relu(want), O.relu(want))
        for c, p in zip((a, b), bufs):
            got = np.empty_like(x)
            L.check(lib.rn_memcpy_d2h(c.handle, got.ctypes.data, p, x.nbytes), "d2h", c.handle)
            assert np.array_equal(got, want)
            L.check(lib.rn_free(c.handle, p), "free", c.handle)
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("case", [(3, 32, 64, 14, 14, 3, 1, 1),      # plain
                                  (5, 64, 72, 7, 7, 1, 1, 0),        # 49 pixels per image: quads straddle images, unaligned planes
                                  (2, 128, 40, 9, 11, 3, 2, 1),      # K = 1152: chunked K sum, ragged channels
                                  (1, 3, 64, 32, 32, 7, 2, 3),       # small-Cin stem form
                                  (40, 128, 64, 7, 7, 3, 1, 1),      # chunked, more than one round of tiles: a tail that the NHWC route cuts
                                  # 1x1 / stride 1 with H*W % 4 == 0: the NCHW-native kernel (rn_conv_nchw.hip)
                                  (3, 64, 256, 8, 8, 1, 1, 0),       # 128 x 128 tiles
                                  (2, 256, 64, 14, 14, 1, 1, 0),     # 64 output channels: the 64 x 256 tile; tiles straddle images
                                  (3, 1024, 200, 6, 6, 1, 1, 0),     # K = 1024: chunked sum; ragged channel tile
                                  (5, 2048, 40, 2, 2, 1, 1, 0),      # chunked, 20 pixels in all, fewer channels than a tile
                                  (7, 96, 136, 10, 10, 1, 1, 0),     # three K tiles, ragged everything
                                  (2, 64, 64, 3, 3, 1, 1, 0),        # 9 pixels per image: one dword load per staged pixel
                                  (3, 2048, 512, 7, 7, 1, 1, 0),     # 7x7 planes, chunked (layer4's conv1)
                                  (2, 64, 128, 8, 8, 1, 2, 0),       # stride 2 (the projection shortcuts)
                                  (3, 256, 72, 9, 7, 1, 2, 0),       # stride 2 on odd sizes
                                  (2, 48, 64, 8, 8, 1, 1, 0),        # Cin not a multiple of 32: the direct kernel, another order
                                  (2, 32, 256, 1, 3, 1, 3, 0),       # ONE output pixel per image (found by tests/fuzz/conv_fuzz.py:
                                  (5, 64, 40, 1, 1, 1, 1, 0),        #  the q / HW multiply-high has no form for HW = 1)
                                  (3, 96, 64, 5, 1, 1, 1, 0)])       # one-pixel-wide planes
def test_nchw_route_writes_the_nhwc_routes_bits(case):
    """rn_conv2d_forward on NCHW tensors: a 1x1 / padding-0 convolution runs on the NCHW-native
    kernel (rn_conv_nchw.hip: weights = MFMA rows, pixels = columns, no transpose, no packing), a
    k x k one on its gathering form (large planes, or rn_ctx_set_nchw_taps(2)) or on the route that
    transposes its input and lets the contraction's epilogue write NCHW itself
    (GemmParams::out_nchw).  Same products and the same summation order as the NHWC call -- also
    where the NHWC launch cuts its tail tiles into K chunks and a finishing kernel adds them -- so
    the two layouts must agree bit for bit, and with the oracle."""
    B, Cin, Cout, H, W, k, s, p = case
    x, w = rnd((B, Cin, H, W), 300 + sum(case)), rnd((Cout, Cin, k, k), 301 + sum(case)) / np.sqrt(Cin * k * k)
    a = ops.conv2d(x, w, s, p, "nchw")
    if Cin >= 4:   # (an NHWC call with fewer than four channels takes the direct kernel: another order)
        assert np.array_equal(a, ops.conv2d(x, w, s, p, "nhwc"))
    assert np.array_equal(a, ops.conv2d(x, w, s, p, "nchw"))
    ctx = R.get_ctx()
    ctx.set_nchw_taps(2)   # k x k: the taps gathered from the channel planes instead of the transpose
    try:
        assert np.array_equal(a, ops.conv2d(x, w, s, p, "nchw"))
    finally:
        ctx.set_nchw_taps(1)
    if B * H * W <= 4000:
        want = O.conv2d(x, w, s, p)
        assert np.abs(a - want).max() <= 2e-6 * np.sqrt(Cin * k * k) * float(np.abs(want).max()) + 1e-6


@pytest.mark.parametrize("case", [(3, 20, 20, 64), (1, 8, 8, 64), (2, 56, 56, 128), (5, 13, 9, 128)])
def test_fp32_conv_chain_is_the_two_separate_launches_bit_for_bit(case):
    """rn_conv_chain_forward_dt, fp32 storage: conv3 + bn3 + residual + ReLU of a 64-channel block and
    conv1 + bn1 + ReLU of the next as one launch (weights as v_mfma_f32_32x32x2_f32 operands in
    registers, the residual tile updated in place into the y tile in LDS).  The same k pairs per
    MFMA and the same epilogue expression as conv_gemm_kernel<float>: y and t1 bit for bit against
    the two rn_conv2d_nhwc_forward calls, y within tolerance of the oracle."""
    B, H, W, N1 = case
    seed = 820 + sum(case)
    t2, x = rnd((B, 64, H, W), seed), rnd((B, 256, H, W), seed + 1)
    w3, w1 = rnd((256, 64, 1, 1), seed + 2) / 8.0, rnd((N1, 256, 1, 1), seed + 3) / 16.0
    g = np.random.default_rng(seed + 4)
    sc3, sh3 = g.random(256, dtype=np.float32) + 0.5, g.standard_normal(256, dtype=np.float32)
    sc1, sh1 = g.random(N1, dtype=np.float32) + 0.5, g.standard_normal(N1, dtype=np.float32)
    want_y = ops.conv2d_nhwc_fused(t2, w3, 1, 0, sc3, sh3, x, True)
    want_t1 = ops.conv2d_nhwc_fused(want_y, w1, 1, 0, sc1, sh1, None, True)
    got_y, got_t1 = ops.conv_chain_f32(t2, x, w3, sc3, sh3, w1, sc1, sh1)
    assert np.array_equal(got_y, want_y)
    assert np.array_equal(got_t1, want_t1)
    if B * H * W <= 2000:
        ref = np.maximum(O.conv2d(t2, w3, 1, 0) * sc3[None, :, None, None] + sh3[None, :, None, None] + x, 0)
        assert np.abs(got_y - ref).max() <= 2e-5 * float(np.abs(ref).max()) + 1e-6


@pytest.mark.parametrize("case", [(3, 20, 20), (2, 56, 56), (1, 9, 7)])
def test_fp32_conv_chain_out_of_the_fused_pair_bit_for_bit(case):
    """rn_conv_chain_pair_forward_dt, fp32: the fused conv3 + downsample pair as the chain's first
    product (conv1's panel as an operand image in LDS), against rn_conv2d_nhwc_pair_forward_dt
    followed by the next block's conv1."""
    B, H, W = case
    seed = 860 + sum(case)
    t2, x2 = rnd((B, 64, H, W), seed), rnd((B, 64, H, W), seed + 1)
    w3, wd = rnd((256, 64, 1, 1), seed + 2) / 8.0, rnd((256, 64, 1, 1), seed + 3) / 8.0
    w1 = rnd((64, 256, 1, 1), seed + 4) / 16.0
    g = np.random.default_rng(seed + 5)
    sc3, scd = g.random(256, dtype=np.float32) + 0.5, g.random(256, dtype=np.float32) + 0.5
    shift = g.standard_normal(256, dtype=np.float32)
    sc1, sh1 = g.random(64, dtype=np.float32) + 0.5, g.standard_normal(64, dtype=np.float32)
    want_y = ops.conv2d_nhwc_pair(t2, w3, x2, wd, 1, 0, 1, sc3, scd, shift, None, True)
    want_t1 = ops.conv2d_nhwc_fused(want_y, w1, 1, 0, sc1, sh1, None, True)
    got_y, got_t1 = ops.conv_chain_pair(t2, x2, w3, sc3, wd, scd, shift, w1, sc1, sh1, bf16=False)
    assert np.array_equal(got_y, want_y)
    assert np.array_equal(got_t1, want_t1)


@pytest.mark.parametrize("shape", [(2, 3, 224, 224), (3, 3, 32, 32), (1, 3, 40, 48), (2, 2, 26, 16),
                                   (1, 3, 256, 256),    # the widest image the launch takes (conv output width 128)
                                   (300, 3, 64, 32)])   # more blocks than CUs; segments that start inside an image
@pytest.mark.parametrize("bf16", [False, True])
def test_fused_stem_and_maxpool_match_the_four_reference_ops(shape, bf16):
    """rn_stem_pool_forward_dt: conv 7x7/2 + batch-norm (folded) + ReLU + max-pool 3x3/2/1 as one
    launch (main.cu:179-192) against the oracle's four ops.  Odd row counts, the first pooled row
    (whose top window row does not exist), windows over the right and bottom edges, fewer than
    three input channels, images cut into segments (a block then computes the stem row above its
    segment once more), the pooled row that two consecutive items of a block share; bf16: oracle on
    bf16-rounded operands, stem output rounded to bf16 before the pool as the unfused path does."""
    B, Cin, H, W = shape
    seed = 1200 + sum(shape)
    x, w = rnd(shape, seed), rnd((64, Cin, 7, 7), seed + 1) / np.sqrt(Cin * 49)
    g = np.random.default_rng(seed + 2)
    sc, sh = g.random(64, dtype=np.float32) + 0.5, g.standard_normal(64, dtype=np.float32) * 0.3
    if bf16:
        x, w = ops.bf16_round(x), ops.bf16_round(w)
    y = O.conv2d(x, w, 2, 3)
    y = O.relu_(y * sc[None, :, None, None] + sh[None, :, None, None])
    if bf16:
        y = ops.bf16_round(y)
    want = O.maxpool2d(y, 3, 2, 1)
    got = ops.stem_pool(x, w, sc, sh, True, bf16=bf16)
    assert got.shape == want.shape
    tol = (2 ** -7 if bf16 else 2e-6 * np.sqrt(Cin * 49)) * float(np.abs(want).max()) + 1e-6
    assert np.abs(got - want).max() <= tol
    assert (got >= 0).all() and np.array_equal(got, ops.stem_pool(x, w, sc, sh, True, bf16=bf16))
    # rn_stem_pool_nchw_forward_dt assembles the same patch from the NCHW image: the same bits
    assert np.array_equal(got, ops.stem_pool(x, w, sc, sh, True, bf16=bf16, from_nchw=True))
    # against the engine's own unfused kernels: same products, another summation order
    if not bf16 and Cin == 3:
        conv = ops.conv2d_nhwc_fused(x, w, 2, 3, sc, sh, None, True)
        assert np.abs(got - ops.maxpool2d(conv, 3, 2, 1, "nhwc")).max() <= tol


@pytest.mark.parametrize("bf16", [False, True])
def test_fused_stem_segments_change_blocks_not_results(bf16):
    """A block of the fused stem walks a segment of an image and carries the pooled row two
    consecutive items share in LDS; a segment that starts inside an image recomputes the stem row
    above it.  Whatever the segment length (rn_ctx_set_stem_items), the bits are the same -- odd
    row counts and the last, partial item included."""
    from resnet_c_amd import _lib as L
    ctx, lib = R.get_ctx(), L.lib()
    for shape in ((2, 3, 224, 224), (3, 3, 90, 48), (1, 2, 58, 64)):
        x, w = rnd(shape, 77 + shape[2]), rnd((64, shape[1], 7, 7), 78) / np.sqrt(shape[1] * 49)
        sc, sh = np.linspace(0.5, 1.5, 64, dtype=np.float32), np.linspace(-0.3, 0.3, 64, dtype=np.float32)
        if bf16:
            x, w = ops.bf16_round(x), ops.bf16_round(w)
        try:
            outs = []
            for items in (1, 2, 3, 5, 100, 0):
                L.check(lib.rn_ctx_set_stem_items(ctx.handle, items), "items", ctx.handle)
                outs.append(ops.stem_pool(x, w, sc, sh, True, bf16=bf16))
                outs.append(ops.stem_pool(x, w, sc, sh, True, bf16=bf16, from_nchw=True))
        finally:
            lib.rn_ctx_set_stem_items(ctx.handle, 0)
        for o in outs[1:]:
            assert np.array_equal(o, outs[0])
        y = O.relu_(O.conv2d(x, w, 2, 3) * sc[None, :, None, None] + sh[None, :, None, None])
        want = O.maxpool2d(ops.bf16_round(y) if bf16 else y, 3, 2, 1)
        assert np.abs(outs[0] - want).max() <= (2 ** -7 if bf16 else 3e-5) * float(np.abs(want).max()) + 1e-6
    assert lib.rn_ctx_set_stem_items(ctx.handle, -1) == L.RN_ERR_INVALID


@pytest.mark.parametrize("seed", range(10))
def test_fused_stem_random_geometries(seed):
    """Random image sizes (odd and even heights; both widths behind a conv output width of 8 .. 64 in
    fp32 -- bf16 rows must be whole 16-byte pieces: even widths), channel counts, batch sizes and segment
    lengths, both input forms, against the oracle's four ops."""
    from resnet_c_amd import _lib as L
    g = np.random.default_rng(4242 + seed)
    bf16 = bool(seed & 1)
    wo = 8 * int(g.integers(1, 9))
    W = 2 * wo - (0 if bf16 else int(g.integers(0, 2)))
    H = int(g.integers(7, 100))
    B, Cin = int(g.integers(1, 5)), int(g.integers(1, 4))
    x, w = rnd((B, Cin, H, W), 5000 + seed), rnd((64, Cin, 7, 7), 5100 + seed) / np.sqrt(Cin * 49)
    sc, sh = g.random(64, dtype=np.float32) + 0.5, g.standard_normal(64, dtype=np.float32) * 0.3
    if bf16:
        x, w = ops.bf16_round(x), ops.bf16_round(w)
    y = O.relu_(O.conv2d(x, w, 2, 3) * sc[None, :, None, None] + sh[None, :, None, None])
    want = O.maxpool2d(ops.bf16_round(y) if bf16 else y, 3, 2, 1)
    ctx, lib = R.get_ctx(), L.lib()
    try:
        for items in (0, int(g.integers(1, 9))):
            L.check(lib.rn_ctx_set_stem_items(ctx.handle, items), "items", ctx.handle)
            for nchw in ((False, True) if W % 4 == 0 else (False,)):
                got = ops.stem_pool(x, w, sc, sh, True, bf16=bf16, from_nchw=nchw)
                assert got.shape == want.shape, (B, Cin, H, W)
                tol = (2 ** -7 if bf16 else 2e-6 * np.sqrt(Cin * 49)) * float(np.abs(want).max()) + 1e-6
                assert np.abs(got - want).max() <= tol, (B, Cin, H, W, items, nchw)
    finally:
        lib.rn_ctx_set_stem_items(ctx.handle, 0)


def test_fused_stem_refuses_what_it_cannot_do():
    from resnet_c_amd import _lib as L
    x, w = rnd((1, 3, 30, 30), 5), rnd((64, 3, 7, 7), 6)
    with pytest.raises(L.RnError):      # conv output width 15: not a multiple of 8
        ops.stem_pool(x, w)
    with pytest.raises(L.RnError):
        ops.stem_pool(x, w, from_nchw=True)
    # without the ReLU the integer maximum of the pool would be wrong for negative values: refused
    x = rnd((1, 3, 32, 32), 7)
    for nchw in (False, True):
        for bf16 in (False, True):
            with pytest.raises(L.RnError) as e:
                ops.stem_pool(x, w, None, None, False, bf16=bf16, from_nchw=nchw)
            assert e.value.status == L.RN_ERR_INVALID and "ReLU" in str(e.value)


@pytest.mark.parametrize("case", [(2, 3, 224, 224), (3, 3, 32, 32), (1, 2, 23, 48), (5, 1, 9, 16), (2, 3, 61, 192)])
def test_fused_stem_that_also_writes_the_stem_tensor(case):
    """rn_stem_conv_pool_nchw_forward (stem_pool_kernel<float, NCHW, WRITE_Y>): the pooled tensor bit for bit what
    rn_stem_pool_nchw_forward_dt gives (the same kernel body), the stem tensor -- every element written: the buffer
    starts as NaNs -- within tolerance of the oracle's conv + affine + ReLU, and the pooled tensor exactly the
    oracle's max-pool of the stem tensor that was written (it is the maximum of those very values)."""
    B, Cin, H, W = case
    x, w = rnd((B, Cin, H, W), 7300 + H), rnd((64, Cin, 7, 7), 7301 + H) / np.sqrt(49 * Cin)
    sc, sh = np.random.default_rng(7302 + H).random(64, dtype=np.float32) + 0.5, rnd((64,), 7303 + H) * 0.2
    y, pooled = ops.stem_conv_pool(x, w, sc, sh)
    assert np.array_equal(pooled, ops.stem_pool(x, w, sc, sh, True, from_nchw=True))
    want = np.maximum(O.conv2d(x, w, 2, 3) * sc[None, :, None, None] + sh[None, :, None, None], 0)
    assert not np.isnan(y).any() and y.shape == want.shape
    assert np.abs(y - want).max() <= 3e-6 * np.sqrt(49 * Cin) * float(np.abs(want).max()) + 1e-5
    assert np.array_equal(pooled, O.maxpool2d(y, 3, 2, 1))


@pytest.mark.parametrize("seed", range(8))
def test_nchw_native_1x1_random_shapes(seed):
    """rn_conv_nchw.hip on random shapes: any batch, plane size (quads that straddle images, planes
    that are not 16-byte multiples), channel counts around the tile edges, strides 1-3, K below and
    above the chunked-sum threshold -- bit for bit the NHWC contraction, and within tolerance of the
    oracle."""
    g = np.random.default_rng(777 + seed)
    B, H, W = int(g.integers(1, 6)), int(g.integers(1, 20)), int(g.integers(1, 20))
    Cin = 32 * int(g.choice([1, 2, 3, 8, 16, 32, 40]))
    Cout = int(g.choice([1, 7, 63, 64, 65, 127, 128, 129, 200, 300]))
    stride = int(g.choice([1, 1, 2, 3]))
    x, w = rnd((B, Cin, H, W), 6000 + seed), rnd((Cout, Cin, 1, 1), 6100 + seed) / np.sqrt(Cin)
    a = ops.conv2d(x, w, stride, 0, "nchw")
    assert np.array_equal(a, ops.conv2d(x, w, stride, 0, "nhwc")), (B, Cin, Cout, H, W, stride)
    want = O.conv2d(x, w, stride, 0)
    assert np.abs(a - want).max() <= 2e-6 * np.sqrt(Cin) * float(np.abs(want).max()) + 1e-6


@pytest.mark.parametrize("seed", range(12))
def test_nchw_native_kxk_random_shapes(seed):
    """rn_conv_nchw.hip, k x k: tap (kh, kw) of a K tile gathered from the channel planes at
    (s oh + kh - pad, s ow + kw - pad), zero in the padding; the packed panel as the MFMA rows.  Kernel sizes
    1-7, paddings from none to more than the kernel needs, strides 1-3, planes down to one pixel, tiles that
    straddle images, K below and above the chunked-sum threshold -- bit for bit the NHWC contraction (same K
    order: taps outermost, 32 channels per tile), and within tolerance of the oracle."""
    g = np.random.default_rng(4242 + seed)
    k = int(g.choice([1, 3, 3, 3, 5, 7]))
    pad = int(g.integers(0, k + 1)) if k > 1 else int(g.integers(0, 3))
    stride = int(g.choice([1, 1, 2, 3]))
    B = int(g.integers(1, 6))
    H, W = int(g.integers(max(1, k - 2 * pad), 20)), int(g.integers(max(1, k - 2 * pad), 20))
    Cin = 32 * int(g.choice([1, 2, 4, 8] if k > 3 else [1, 2, 3, 4, 8, 16]))
    Cout = int(g.choice([1, 7, 63, 64, 65, 127, 128, 129, 200]))
    x, w = rnd((B, Cin, H, W), 6200 + seed), rnd((Cout, Cin, k, k), 6300 + seed) / np.sqrt(Cin * k * k)
    ctx = R.get_ctx()
    ctx.set_nchw_taps(2)
    try:
        a = ops.conv2d(x, w, stride, pad, "nchw")
    finally:
        ctx.set_nchw_taps(1)
    assert np.array_equal(a, ops.conv2d(x, w, stride, pad, "nhwc")), (B, Cin, Cout, H, W, k, stride, pad)
    want = O.conv2d(x, w, stride, pad)
    assert a.shape == want.shape
    assert np.abs(a - want).max() <= 2e-6 * np.sqrt(Cin * k * k) * float(np.abs(want).max()) + 1e-6


def test_nchw_native_kxk_is_a_choice_of_route_not_of_bits():
    """rn_ctx_set_nchw_taps: 0 keeps the transposing route for every k x k layer on NCHW tensors, 2 takes the
    gathering kernel wherever it is eligible, 1 (the default) takes it for planes of 2048 pixels or more (where it
    measures faster, tools/nchw_bench.py): the same bits whichever runs, on a small plane and on a large one."""
    ctx = R.get_ctx()
    try:
        for (B, Cin, Cout, H, W, s_) in ((3, 64, 96, 14, 14, 1), (2, 64, 64, 56, 56, 1), (2, 128, 128, 56, 56, 2)):
            x, w = rnd((B, Cin, H, W), 6400 + H), rnd((Cout, Cin, 3, 3), 6401 + H) / np.sqrt(9 * Cin)
            outs = []
            for mode in (2, 1, 0):
                ctx.set_nchw_taps(mode)
                outs.append(ops.conv2d(x, w, s_, 1, "nchw"))
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
            assert np.array_equal(outs[0], ops.conv2d(x, w, s_, 1, "nhwc"))
        with pytest.raises(R.RnError):
            ctx.set_nchw_taps(3)
    finally:
        ctx.set_nchw_taps(1)


@pytest.mark.parametrize("case", [(16, 7, 7, 512, 2048, 1, 1, 0),     # layer4 conv3: the shape the per-launch rule groups
                                  (9, 14, 14, 256, 256, 3, 1, 1),     # chunked K sum with a cut tail behind the remap
                                  (5, 9, 11, 64, 320, 1, 1, 0),       # ragged: 5 N tiles (no grouping possible), ragged M
                                  (7, 13, 13, 96, 256, 3, 2, 1),      # whole rows + a partial last row of M panels
                                  (3, 28, 28, 128, 512, 1, 1, 0)])
def test_xcd_tile_order_changes_blocks_not_bits(case):
    """rn_ctx_set_xcd_groups: the tiles are dealt to the XCDs in 1 / 2 / 4 / 8 groups of N tiles (a bijection of
    the tile index for any tile count: whole rows of M panels are reordered, the tiles past them keep their
    place).  Every order, on every 4-wave tile candidate, writes the bits of the logical order; against the oracle
    once."""
    from resnet_c_amd import _lib as L
    B, H, W, Cin, Cout, k, s, p = case
    x, w = rnd((B, Cin, H, W), 40 + sum(case)), rnd((Cout, Cin, k, k), 41 + sum(case)) / np.sqrt(Cin * k * k)
    g = np.random.default_rng(42 + sum(case))
    sc, sh = g.random(Cout, dtype=np.float32) + 0.5, g.standard_normal(Cout, dtype=np.float32)
    ctx, lib = R.get_ctx(), L.lib()
    try:
        ctx.set_xcd_groups(1)
        lib.rn_ctx_set_conv_tile(ctx.handle, 4)
        want = ops.conv2d_nhwc_fused(x, w, s, p, sc, sh, None, True)
        for groups in (0, 1, 2, 4, 8):
            ctx.set_xcd_groups(groups)
            for cand in (1, 2, 3, 4, 6, 8):
                lib.rn_ctx_set_conv_tile(ctx.handle, cand)
                got = ops.conv2d_nhwc_fused(x, w, s, p, sc, sh, None, True)
                assert np.array_equal(got, want), (groups, cand)
    finally:
        ctx.set_xcd_groups(0)
        lib.rn_ctx_set_conv_tile(ctx.handle, 0)
    y = O.conv2d(x, w, s, p)
    ref = np.maximum(y * sc[None, :, None, None] + sh[None, :, None, None], 0)
    assert np.abs(want - ref).max() <= 3e-6 * np.sqrt(Cin * k * k) * float(np.abs(ref).max()) + 1e-5
    with pytest.raises(L.RnError):
        ctx.set_xcd_groups(3)

