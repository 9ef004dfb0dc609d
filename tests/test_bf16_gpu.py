"""bf16 storage / fp32 accumulate (BASELINE.json configs[3]).  The reference is fp32 only;
the oracle for these kernels is the same CPU restatement run on bf16-rounded operands:
products of bf16 values are exact in fp32, so the only differences are fp32 summation
order and the final rounding of the output to bf16 (rel. 2^-9)."""
import ctypes
import os

import numpy as np
import pytest

import resnet_c_amd as R
from oracle import oracle as O
from resnet_c_amd import ops

pytestmark = pytest.mark.gpu


def rnd(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def test_bf16_rounding_helper_is_rne():
    x = np.array([1.0, 1.00390625, 1.005859375, -2.5, 3.1415926, 0.0, 65504.0], dtype=np.float32)
    b = ops.bf16_round(x)
    assert b[0] == 1.0 and b[1] == 1.0  # 1 + 2^-8 is a tie -> even mantissa
    assert b[2] == 1.0078125 and b[3] == -2.5
    assert abs(b[4] - 3.1415926) <= 2 ** -7 and b[5] == 0.0


CASES = [(2, 64, 64, 12, 12, 1, 1, 0), (1, 128, 128, 10, 10, 3, 1, 1), (2, 64, 96, 13, 11, 3, 2, 1),
         (1, 256, 512, 8, 8, 1, 2, 0), (1, 3, 64, 32, 32, 7, 2, 3), (1, 3, 64, 224, 224, 7, 2, 3),
         (3, 2048, 1000, 1, 1, 1, 1, 0)]


@pytest.mark.parametrize("case", CASES)
def test_bf16_conv_matches_oracle_on_rounded_operands(case):
    B, Cin, Cout, H, W, k, s, p = case
    x, w = rnd((B, Cin, H, W), 7 + sum(case)), rnd((Cout, Cin, k, k), 8 + sum(case)) / np.sqrt(Cin * k * k)
    want = O.conv2d(ops.bf16_round(x), ops.bf16_round(w), s, p)
    got32 = ops.conv2d_nhwc_bf16(x, w, s, p, out_f32=True)
    scale = float(np.abs(want).max())
    assert np.abs(got32 - want).max() <= 3e-7 * np.sqrt(Cin * k * k) * scale + 1e-6
    got16 = ops.conv2d_nhwc_bf16(x, w, s, p)
    if Cin * k * k < 32 * 64:
        assert np.array_equal(got16, ops.bf16_round(got32))  # bf16 output = RNE of the fp32 result
    else:
        # 32 K tiles and more: the fp32-result form (a bf16 model's fc) adds K in eight chunks, the bf16-result
        # form in one run -- two fp32 sums of the same products, whose roundings to bf16 may differ by one step
        step = np.spacing(np.abs(got16).astype(np.float32)) * 2.0 ** 16
        assert (np.abs(got16 - ops.bf16_round(got32)) <= step).all()
        assert (got16 != ops.bf16_round(got32)).mean() < 0.02


WIDE_CASES = [(3, 64, 64, 20, 20, 3, 1, 1),      # 1200 rows: ragged last M tile, padded taps
              (2, 128, 256, 17, 15, 3, 2, 1),    # stride 2, odd image, two channel segments per tap
              (5, 256, 96, 9, 9, 1, 1, 0),       # N tile wider than Cout
              (2, 512, 520, 8, 8, 1, 2, 0),      # ragged N tile, strided 1x1
              (1, 64, 64, 6, 6, 3, 1, 1)]        # one K tile per tap, a single ragged tile


@pytest.mark.parametrize("case", WIDE_CASES)
def test_bf16_every_tile_candidate_matches_oracle_and_each_other(case):
    """Candidates 1-8 are the 4-wave kernel's tiles, 9.. the 8-wave LDS-DMA kernel's 256-wide
    tiles (rn_conv_wide.hip): each against the oracle on bf16-rounded operands (padded taps,
    rows past M and channels past Cout must come out of the DMA as zeros), and all with the
    same bits -- the k order per output element does not depend on the tile."""
    from resnet_c_amd import _lib as L
    B, Cin, Cout, H, W, k, s, p = case
    x, w = rnd((B, Cin, H, W), 107 + sum(case)), rnd((Cout, Cin, k, k), 108 + sum(case)) / np.sqrt(Cin * k * k)
    g = np.random.default_rng(109 + sum(case))
    scale, shift = g.random(Cout, dtype=np.float32) + 0.5, g.standard_normal(Cout, dtype=np.float32)
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rnd((B, Cout, ho, wo), 110 + sum(case))
    y = O.conv2d(ops.bf16_round(x), ops.bf16_round(w), s, p)
    want = np.maximum(y * scale[None, :, None, None] + shift[None, :, None, None] + ops.bf16_round(res), 0)
    ctx, lib = R.get_ctx(), L.lib()
    base = ops.conv2d_nhwc_bf16(x, w, s, p, scale, shift, res, True)
    assert np.abs(base - want).max() <= 2 ** -8 * np.abs(want).max() + 1e-5
    plain = ops.conv2d_nhwc_bf16(x, w, s, p)
    try:
        for cand in range(1, lib.rn_conv_tile_candidates() + 1):
            lib.rn_ctx_set_conv_tile(ctx.handle, cand)
            assert np.array_equal(ops.conv2d_nhwc_bf16(x, w, s, p, scale, shift, res, True), base), cand
            assert np.array_equal(ops.conv2d_nhwc_bf16(x, w, s, p), plain), cand
    finally:
        lib.rn_ctx_set_conv_tile(ctx.handle, 0)


def test_bf16_conv_pair_every_tile_candidate():
    from resnet_c_amd import _lib as L
    B, Cin, Cout, H, W, Cin2, s2 = 3, 128, 256, 10, 10, 64, 2
    t, w = rnd((B, Cin, H, W), 171), rnd((Cout, Cin, 1, 1), 172)
    x2, w2 = rnd((B, Cin2, 19, 19), 173), rnd((Cout, Cin2, 1, 1), 174)
    g = np.random.default_rng(175)
    sc1, sc2 = g.random(Cout, dtype=np.float32) + 0.5, g.random(Cout, dtype=np.float32) + 0.5
    shift = g.standard_normal(Cout, dtype=np.float32)
    ctx, lib = R.get_ctx(), L.lib()
    base = ops.conv2d_nhwc_pair(t, w, x2, w2, 1, 0, s2, sc1, sc2, shift, None, True, bf16=True)
    tb, xb = ops.bf16_round(t), ops.bf16_round(x2)
    w1b = ops.bf16_round(w * sc1[:, None, None, None])
    w2b = ops.bf16_round(w2 * sc2[:, None, None, None])
    want = O.relu_(O.conv2d(tb, w1b, 1, 0) + O.conv2d(xb, w2b, s2, 0) + shift[None, :, None, None])
    np.testing.assert_allclose(base, ops.bf16_round(want), rtol=2 ** -7, atol=2e-2)
    try:
        for cand in range(1, lib.rn_conv_tile_candidates() + 1):
            lib.rn_ctx_set_conv_tile(ctx.handle, cand)
            got = ops.conv2d_nhwc_pair(t, w, x2, w2, 1, 0, s2, sc1, sc2, shift, None, True, bf16=True)
            assert np.array_equal(got, base), cand
    finally:
        lib.rn_ctx_set_conv_tile(ctx.handle, 0)


def test_wide_kernel_race_screen_at_full_size():
    """The LDS-DMA ring is ordered by counted vmcnt waits and one barrier per K step.  A read
    that slipped ahead of its DMA would show as rare wrong tiles that come and go with timing, so
    this runs the real layer shapes (B=256: 196 blocks of 256x256 on layer3's 3x3, 196 of 256x128
    on layer4's) many times under load on device-resident buffers and holds every run to the
    4-wave kernel's bits."""
    from resnet_c_amd import _lib as L
    from resnet_c_amd.tensor import _DeviceBuffer
    ctx, lib = R.get_ctx(), L.lib()
    for (B, H, W, Cin, Cout, k, pad, cands) in [(256, 14, 14, 256, 256, 3, 1, (9, 10, 11)),
                                                (256, 7, 7, 512, 512, 3, 1, (9, 10, 11)),
                                                (64, 28, 28, 128, 128, 3, 1, (10, 12)),
                                                (256, 14, 14, 1024, 256, 1, 0, (9,))]:
        g = np.random.default_rng(B + Cin + k)
        n_in, n_w, n_out = B * H * W * Cin, Cout * k * k * Cin, B * H * W * Cout
        x = _DeviceBuffer(ctx, n_in * 2)
        w = _DeviceBuffer(ctx, n_w * 2)
        xh = ops.to_bf16_bits(g.standard_normal(n_in, dtype=np.float32))
        wh = ops.to_bf16_bits(g.standard_normal(n_w, dtype=np.float32) / np.sqrt(Cin * k * k))
        L.check(lib.rn_memcpy_h2d(ctx.handle, x.ptr, xh.ctypes.data, xh.nbytes), "h2d", ctx.handle)
        L.check(lib.rn_memcpy_h2d(ctx.handle, w.ptr, wh.ctypes.data, wh.nbytes), "h2d", ctx.handle)
        out = _DeviceBuffer(ctx, n_out * 2)
        ep = L.Epilogue(None, None, None, 1)

        def run(cand):
            lib.rn_ctx_set_conv_tile(ctx.handle, cand)
            L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, L.RN_DTYPE_BF16, L.RN_DTYPE_BF16, x.ptr, out.ptr, w.ptr,
                                                  k, 1, pad, H, W, B, Cin, Cout, H, W, ctypes.byref(ep)), "conv", ctx.handle)

        def fetch():
            ctx.sync()
            h = np.empty(n_out, dtype=np.uint16)
            L.check(lib.rn_memcpy_d2h(ctx.handle, h.ctypes.data, out.ptr, h.nbytes), "d2h", ctx.handle)
            return h

        try:
            run(5)
            want = fetch()
            assert want.any()
            for cand in cands:
                for rep in range(12):
                    for _ in range(6):      # back-to-back launches: the chip is warm and loaded
                        run(cand)
                    assert np.array_equal(fetch(), want), (cand, rep)
        finally:
            lib.rn_ctx_set_conv_tile(ctx.handle, 0)


STRIP_CASES = [(3, 20, 20, 64), (1, 6, 6, 64), (2, 56, 56, 64), (5, 7, 61, 64), (4, 1, 9, 64), (2, 33, 5, 64),
               (37, 28, 28, 64), (3, 12, 12, 128), (1, 6, 6, 128), (2, 28, 28, 128), (5, 7, 29, 128), (4, 1, 9, 128),
               (41, 14, 14, 128)]


@pytest.mark.parametrize("case", STRIP_CASES)
def test_strip_kernel_matches_oracle_and_the_tile_kernels_bits(case):
    """conv_strip_kernel / conv_strip128_kernel (3x3 / stride 1 / 64 -> 64 and 128 -> 128 channels:
    weights in registers, the zero-padded image in a rolling LDS ring, operands swapped, stores straight
    from registers) -- the candidate after the wide tiles.  Against the oracle on bf16-rounded operands
    and bit for bit against a 4-wave tile: widest image the ring margin allows (61 / 29), one-row and
    narrow images, several images per step and several steps per block (37 x 28 x 28 x 64: 479 steps
    on 256 blocks; 41 x 14 x 14 x 128: 308 steps)."""
    from resnet_c_amd import _lib as L
    B, H, W, C = case
    x, w = rnd((B, C, H, W), 900 + sum(case)), rnd((C, C, 3, 3), 901 + sum(case)) / np.sqrt(9.0 * C)
    g = np.random.default_rng(902 + sum(case))
    sc, sh = g.random(C, dtype=np.float32) + 0.5, g.standard_normal(C, dtype=np.float32)
    ctx, lib = R.get_ctx(), L.lib()
    strip = lib.rn_conv_tile_candidates()
    try:
        lib.rn_ctx_set_conv_tile(ctx.handle, 4)
        want_ep = ops.conv2d_nhwc_bf16(x, w, 1, 1, sc, sh, None, True)
        want_plain = ops.conv2d_nhwc_bf16(x, w, 1, 1, out_f32=False)
        lib.rn_ctx_set_conv_tile(ctx.handle, strip)
        got_ep = ops.conv2d_nhwc_bf16(x, w, 1, 1, sc, sh, None, True)
        got_plain = ops.conv2d_nhwc_bf16(x, w, 1, 1, out_f32=False)
    finally:
        lib.rn_ctx_set_conv_tile(ctx.handle, 0)
    assert np.array_equal(got_ep, want_ep) and np.array_equal(got_plain, want_plain)
    if B * H * W <= 8000:
        y = O.conv2d(ops.bf16_round(x), ops.bf16_round(w), 1, 1)
        want = np.maximum(y * sc[None, :, None, None] + sh[None, :, None, None], 0)
        assert np.abs(got_ep - want).max() <= 2 ** -8 * np.abs(want).max() + 1e-5


def test_strip_kernel_full_size_under_load():
    """B = 256 at 56 x 56: 3,307 steps, 13 per block -- the ring wraps and every slot is rewritten
    several times while the neighbours are still being read.  Many back-to-back runs against the
    4-wave kernel's bits (a read that slipped ahead of its DMA would come and go with timing)."""
    from resnet_c_amd import _lib as L
    from resnet_c_amd.tensor import _DeviceBuffer
    ctx, lib = R.get_ctx(), L.lib()
    B, H, W, C = 256, 56, 56, 64
    g = np.random.default_rng(77)
    n_in, n_w = B * H * W * C, C * 9 * C
    x, w, out = _DeviceBuffer(ctx, n_in * 2), _DeviceBuffer(ctx, n_w * 2), _DeviceBuffer(ctx, n_in * 2)
    xh = ops.to_bf16_bits(g.standard_normal(n_in, dtype=np.float32))
    wh = ops.to_bf16_bits(g.standard_normal(n_w, dtype=np.float32) / 24.0)
    L.check(lib.rn_memcpy_h2d(ctx.handle, x.ptr, xh.ctypes.data, xh.nbytes), "h2d", ctx.handle)
    L.check(lib.rn_memcpy_h2d(ctx.handle, w.ptr, wh.ctypes.data, wh.nbytes), "h2d", ctx.handle)
    ep = L.Epilogue(None, None, None, 1)

    def run(cand):
        lib.rn_ctx_set_conv_tile(ctx.handle, cand)
        L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, L.RN_DTYPE_BF16, L.RN_DTYPE_BF16, x.ptr, out.ptr, w.ptr,
                                              3, 1, 1, H, W, B, C, C, H, W, ctypes.byref(ep)), "conv", ctx.handle)

    def fetch():
        ctx.sync()
        h = np.empty(n_in, dtype=np.uint16)
        L.check(lib.rn_memcpy_d2h(ctx.handle, h.ctypes.data, out.ptr, h.nbytes), "d2h", ctx.handle)
        return h

    try:
        run(6)
        want = fetch()
        assert want.any()
        L.check(lib.rn_memset(ctx.handle, out.ptr, 0xFF, n_in * 2), "memset", ctx.handle)
        for rep in range(8):
            for _ in range(5):
                run(lib.rn_conv_tile_candidates())
            assert np.array_equal(fetch(), want), rep
    finally:
        lib.rn_ctx_set_conv_tile(ctx.handle, 0)


@pytest.mark.parametrize("case", [(3, 20, 20, 64, 64), (1, 8, 8, 64, 64), (2, 56, 56, 64, 128), (5, 13, 9, 64, 128),
                                  (2, 28, 28, 128, 128), (3, 9, 7, 128, 128), (1, 4, 8, 128, 128)])
def test_conv_chain_is_the_two_separate_launches_bit_for_bit(case):
    """rn_conv_chain_forward_dt (conv3 + bn3 + residual + ReLU of a block, conv1 + bn1 + ReLU of the
    next, y through LDS) against the two rn_conv2d_nhwc_forward_dt calls it replaces: y and t1 bit
    for bit -- same k order, same epilogue expression, y rounded to bf16 before conv1 multiplies
    it -- and y against the oracle.  64 mid channels (8 waves, 64-row steps; both widths of the
    next block's conv1) and 128 (4 waves with 448 registers each, 32-row steps); ragged last
    steps (1200, 585 and 189 rows), one-step launches."""
    B, H, W, MID, N1 = case
    C = 4 * MID
    seed = 700 + sum(case)
    t2, x = rnd((B, MID, H, W), seed), rnd((B, C, H, W), seed + 1)
    w3, w1 = rnd((C, MID, 1, 1), seed + 2) / np.sqrt(MID), rnd((N1, C, 1, 1), seed + 3) / np.sqrt(C)
    g = np.random.default_rng(seed + 4)
    sc3, sh3 = g.random(C, dtype=np.float32) + 0.5, g.standard_normal(C, dtype=np.float32)
    sc1, sh1 = g.random(N1, dtype=np.float32) + 0.5, g.standard_normal(N1, dtype=np.float32)
    want_y = ops.conv2d_nhwc_bf16(t2, w3, 1, 0, sc3, sh3, x, True)
    want_t1 = ops.conv2d_nhwc_bf16(want_y, w1, 1, 0, sc1, sh1, None, True)
    got_y, got_t1 = ops.conv_chain_bf16(t2, x, w3, sc3, sh3, w1, sc1, sh1)
    assert np.array_equal(got_y, want_y)
    assert np.array_equal(got_t1, want_t1)
    if B * H * W <= 2000:
        y = O.conv2d(ops.bf16_round(t2), ops.bf16_round(w3), 1, 0)
        ref = np.maximum(y * sc3[None, :, None, None] + sh3[None, :, None, None] + ops.bf16_round(x), 0)
        assert np.abs(got_y - ref).max() <= 2 ** -8 * np.abs(ref).max() + 1e-5


@pytest.mark.parametrize("case", [(3, 20, 20, 64), (2, 56, 56, 64), (1, 9, 7, 128)])
def test_conv_chain_out_of_the_fused_pair_bit_for_bit(case):
    """rn_conv_chain_pair_forward_dt: the fused conv3 + downsample pair (K = 64 + 64 from two
    tensors, scales folded into the panel, no residual) as the chain's first product, against
    rn_conv2d_nhwc_pair_forward_dt followed by the next block's conv1."""
    B, H, W, N1 = case
    seed = 760 + sum(case)
    t2, x2 = rnd((B, 64, H, W), seed), rnd((B, 64, H, W), seed + 1)
    w3, wd = rnd((256, 64, 1, 1), seed + 2) / 8.0, rnd((256, 64, 1, 1), seed + 3) / 8.0
    w1 = rnd((N1, 256, 1, 1), seed + 4) / 16.0
    g = np.random.default_rng(seed + 5)
    sc3, scd = g.random(256, dtype=np.float32) + 0.5, g.random(256, dtype=np.float32) + 0.5
    shift = g.standard_normal(256, dtype=np.float32)
    sc1, sh1 = g.random(N1, dtype=np.float32) + 0.5, g.standard_normal(N1, dtype=np.float32)
    want_y = ops.conv2d_nhwc_pair(t2, w3, x2, wd, 1, 0, 1, sc3, scd, shift, None, True, bf16=True)
    want_t1 = ops.conv2d_nhwc_bf16(want_y, w1, 1, 0, sc1, sh1, None, True)
    got_y, got_t1 = ops.conv_chain_pair_bf16(t2, x2, w3, sc3, wd, scd, shift, w1, sc1, sh1)
    assert np.array_equal(got_y, want_y)
    assert np.array_equal(got_t1, want_t1)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_chain_kernels_full_size_under_load(dtype):
    """The chained launches order their LDS double buffers with one vmcnt(0) + barrier per step and a
    second barrier before the y tile is read.  A read that slipped ahead of its DMA would show as rare
    wrong rows that come and go with timing: the stage-1 shape at B = 96 (4,704 steps of 64 rows, 18
    per block) many times back to back on device-resident buffers, every run held to the two
    separate launches' bits."""
    from resnet_c_amd import _lib as L
    from resnet_c_amd.tensor import _DeviceBuffer
    ctx, lib = R.get_ctx(), L.lib()
    B, H, W, N1 = 96, 56, 56, 64
    rows = B * H * W
    bf = dtype == "bf16"
    dt, es = (L.RN_DTYPE_BF16, 2) if bf else (L.RN_DTYPE_F32, 4)
    g = np.random.default_rng(5)

    def buf(n, scale):
        h = g.standard_normal(n, dtype=np.float32) * scale
        h = ops.to_bf16_bits(h) if bf else h
        b = _DeviceBuffer(ctx, n * es)
        L.check(lib.rn_memcpy_h2d(ctx.handle, b.ptr, h.ctypes.data, h.nbytes), "h2d", ctx.handle)
        return b

    t2, x = buf(rows * 64, 0.5), buf(rows * 256, 0.5)
    w3, w1 = buf(256 * 64, 0.12), buf(N1 * 256, 0.06)   # 1x1 panels: the packed layout is [Cout][Cin]
    y, t1 = _DeviceBuffer(ctx, rows * 256 * es), _DeviceBuffer(ctx, rows * N1 * es)
    one = R.FloatTensor.from_numpy(np.full(256, 0.75, np.float32), R.Device.GPU)
    ep3, ep1 = L.Epilogue(one.data(), one.data(), x.ptr, 1), L.Epilogue(one.data(), one.data(), None, 1)

    def fetch():
        ctx.sync()
        a, b = np.empty(rows * 256 * es, np.uint8), np.empty(rows * N1 * es, np.uint8)
        L.check(lib.rn_memcpy_d2h(ctx.handle, a.ctypes.data, y.ptr, a.nbytes), "d2h", ctx.handle)
        L.check(lib.rn_memcpy_d2h(ctx.handle, b.ctypes.data, t1.ptr, b.nbytes), "d2h", ctx.handle)
        return a, b

    L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, dt, dt, t2.ptr, y.ptr, w3.ptr, 1, 1, 0, H, W, B, 64, 256, H, W,
                                          ctypes.byref(ep3)), "conv3", ctx.handle)
    L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, dt, dt, y.ptr, t1.ptr, w1.ptr, 1, 1, 0, H, W, B, 256, N1, H, W,
                                          ctypes.byref(ep1)), "conv1", ctx.handle)
    want_y, want_t1 = fetch()
    assert want_t1.any()
    for rep in range(6):
        L.check(lib.rn_memset(ctx.handle, y.ptr, 0xFF, rows * 256 * es), "memset", ctx.handle)
        L.check(lib.rn_memset(ctx.handle, t1.ptr, 0xFF, rows * N1 * es), "memset", ctx.handle)
        for _ in range(5):
            L.check(lib.rn_conv_chain_forward_dt(ctx.handle, dt, t2.ptr, x.ptr, y.ptr, w3.ptr, one.data(), one.data(),
                                                 t1.ptr, w1.ptr, one.data(), one.data(), rows, 64, 256, N1),
                    "chain", ctx.handle)
        got_y, got_t1 = fetch()
        assert np.array_equal(got_y, want_y) and np.array_equal(got_t1, want_t1), rep


@pytest.mark.parametrize("seed", range(6))
def test_wide_kernel_random_shapes_against_the_4_wave_kernel(seed):
    """Random geometry (image size, stride, padding, channel counts that leave ragged M and N
    tiles, 1x1 and 3x3, with and without residual): every wide tile == 4-wave tile, bit for bit."""
    from resnet_c_amd import _lib as L
    g = np.random.default_rng(4000 + seed)
    k = int(g.choice([1, 3]))
    s = int(g.choice([1, 2]))
    p = int(g.choice([0, 1])) if k == 3 else 0
    B, H, W = int(g.integers(1, 9)), int(g.integers(7, 30)), int(g.integers(7, 30))
    Cin, Cout = 64 * int(g.integers(1, 5)), 8 * int(g.integers(1, 50))
    x, w = rnd((B, Cin, H, W), seed), rnd((Cout, Cin, k, k), seed + 1) / np.sqrt(Cin * k * k)
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rnd((B, Cout, ho, wo), seed + 2) if seed % 2 else None
    sc, sh = g.random(Cout, dtype=np.float32) + 0.5, g.standard_normal(Cout, dtype=np.float32)
    ctx, lib = R.get_ctx(), L.lib()
    try:
        lib.rn_ctx_set_conv_tile(ctx.handle, 4)
        want = ops.conv2d_nhwc_bf16(x, w, s, p, sc, sh, res, True)
        for cand in range(9, lib.rn_conv_tile_candidates() + 1):
            lib.rn_ctx_set_conv_tile(ctx.handle, cand)
            assert np.array_equal(ops.conv2d_nhwc_bf16(x, w, s, p, sc, sh, res, True), want), (cand, B, H, W, Cin, Cout, k, s, p)
    finally:
        lib.rn_ctx_set_conv_tile(ctx.handle, 0)


def test_bf16_fused_epilogue():
    B, Cin, Cout, H, W = 2, 64, 128, 9, 9
    x, w = rnd((B, Cin, H, W), 1), rnd((Cout, Cin, 3, 3), 2) / 24
    g = np.random.default_rng(3)
    scale, shift = g.random(Cout, dtype=np.float32) + 0.5, g.standard_normal(Cout, dtype=np.float32)
    res = rnd((B, Cout, H, W), 4)
    y = O.conv2d(ops.bf16_round(x), ops.bf16_round(w), 1, 1)
    want = np.maximum(y * scale[None, :, None, None] + shift[None, :, None, None] + ops.bf16_round(res), 0)
    got = ops.conv2d_nhwc_bf16(x, w, 1, 1, scale, shift, res, True)
    assert np.abs(got - want).max() <= 2 ** -8 * np.abs(want).max() + 1e-5


def test_bf16_conv_pair():
    B, Cin, Cout, H, W, Cin2, s2 = 2, 64, 128, 7, 7, 128, 2
    t, w = rnd((B, Cin, H, W), 71), rnd((Cout, Cin, 1, 1), 72)
    x2, w2 = rnd((B, Cin2, 13, 13), 73), rnd((Cout, Cin2, 1, 1), 74)
    g = np.random.default_rng(75)
    sc1, sc2 = g.random(Cout, dtype=np.float32) + 0.5, g.random(Cout, dtype=np.float32) + 0.5
    shift = g.standard_normal(Cout, dtype=np.float32)
    # what the kernel multiplies: bf16(activations) x bf16(fl32(w * scale)), fp32 accumulate
    tb, xb = ops.bf16_round(t), ops.bf16_round(x2)
    w1b = ops.bf16_round(w * sc1[:, None, None, None])
    w2b = ops.bf16_round(w2 * sc2[:, None, None, None])
    want = O.relu_(O.conv2d(tb, w1b, 1, 0) + O.conv2d(xb, w2b, s2, 0) + shift[None, :, None, None])
    got = ops.conv2d_nhwc_pair(t, w, x2, w2, 1, 0, s2, sc1, sc2, shift, None, True, bf16=True)
    np.testing.assert_allclose(got, ops.bf16_round(want), rtol=2 ** -7, atol=1e-2)


@pytest.mark.parametrize("case", [(2, 3, 224, 224, 4, 3), (2, 3, 9, 10, 4, 2), (1, 3, 6, 7, 4, 1),
                                  (2, 2, 5, 5, 8, 1)])
def test_bf16_bordered_image(case):
    """rn_nchw_to_nhwc_pad_dt(BF16): fp32 NCHW -> bf16 [B,H+2b,W+2b,Cpad], RNE, zero border (the
    two-pixels-per-store kernel for Cpad = 4 and an even padded width, element kernel otherwise)."""
    from resnet_c_amd import _lib as L
    from resnet_c_amd.tensor import _DeviceBuffer
    B, C, H, W, cpad, border = case
    ctx, lib = R.get_ctx(), L.lib()
    x = rnd((B, C, H, W), 40 + sum(case))
    src = R.FloatTensor.from_numpy(x, R.Device.GPU)
    Hp, Wp = H + 2 * border, W + 2 * border
    n = B * Hp * Wp * cpad
    dst = _DeviceBuffer(ctx, n * 2)
    L.check(lib.rn_memset(ctx.handle, dst.ptr, 0xFF, n * 2), "memset", ctx.handle)
    L.check(lib.rn_nchw_to_nhwc_pad_dt(ctx.handle, L.RN_DTYPE_BF16, src.data(), dst.ptr, B, C, H, W,
                                       cpad, border), "pad_dt", ctx.handle)
    ctx.sync()
    got = np.empty(n, dtype=np.uint16)
    L.check(lib.rn_memcpy_d2h(ctx.handle, got.ctypes.data, dst.ptr, got.nbytes), "d2h", ctx.handle)
    want = np.zeros((B, Hp, Wp, cpad), dtype=np.float32)
    want[:, border:border + H, border:border + W, :C] = x.transpose(0, 2, 3, 1)
    assert np.array_equal(got.reshape(B, Hp, Wp, cpad), ops.to_bf16_bits(want).reshape(B, Hp, Wp, cpad))


def test_bf16_pools():
    x = rnd((2, 64, 14, 14), 5)
    xb = ops.bf16_round(x)
    assert np.array_equal(ops.pool_nhwc_bf16(x, 3, 2, 1, True), O.maxpool2d(xb, 3, 2, 1))
    for shape in ((2, 64, 112, 112), (1, 16, 33, 20)):   # eight or more output rows: the column walk
        y = rnd(shape, 7 + shape[2])
        y[0, 1, 4:9, :] = np.nan
        y[0, 2, 0:6, 0:6] = -np.inf
        got = ops.pool_nhwc_bf16(y, 3, 2, 1, True)
        assert np.array_equal(got, O.maxpool2d(ops.bf16_round(y), 3, 2, 1)) and not np.isnan(got).any()
    x7 = rnd((2, 2048, 7, 7), 6)
    want = ops.bf16_round(O.avgpool2d(ops.bf16_round(x7), 7))
    assert np.array_equal(ops.pool_nhwc_bf16(x7, 7, 1, 0, False), want)


def test_bf16_unsupported_shapes_are_reported_not_guessed():
    from resnet_c_amd import _lib as L
    with pytest.raises(R.RnError) as e:
        ops.conv2d_nhwc_bf16(rnd((1, 32, 4, 4), 1), rnd((8, 32, 1, 1), 2))
    assert e.value.status == L.RN_ERR_UNSUPPORTED


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_chain_refuses_misaligned_or_aliased_tensors(dtype):
    """rn_conv_chain_forward_dt moves 16-byte pieces and reads rows ahead of the rows it writes:
    a pointer off a 16-byte boundary or an output on top of an input is a status, not a launch."""
    from resnet_c_amd import _lib as L
    from resnet_c_amd.tensor import _DeviceBuffer
    ctx, lib = R.get_ctx(), L.lib()
    dt, es = (L.RN_DTYPE_BF16, 2) if dtype == "bf16" else (L.RN_DTYPE_F32, 4)
    rows = 128
    t2, x, y, t1 = (_DeviceBuffer(ctx, rows * c * es + 64) for c in (64, 256, 256, 64))
    w3, w1 = _DeviceBuffer(ctx, 64 * 256 * es + 64), _DeviceBuffer(ctx, 256 * 64 * es + 64)
    sc = _DeviceBuffer(ctx, 256 * 4 + 64)

    def call(t2p, xp, yp, t1p, w3p=w3.ptr, scp=None):
        return lib.rn_conv_chain_forward_dt(ctx.handle, dt, t2p, xp, yp, w3p, scp, None, t1p, w1.ptr, None, None,
                                            rows, 64, 256, 64)

    for c in (t2, x, y, t1, w3, w1):
        L.check(lib.rn_memset(ctx.handle, c.ptr, 0, c.nbytes), "memset", ctx.handle)
    assert call(t2.ptr, x.ptr, y.ptr, t1.ptr) == L.RN_OK
    ctx.sync()
    assert call(t2.ptr + 8, x.ptr, y.ptr, t1.ptr) == L.RN_ERR_INVALID      # misaligned input
    assert call(t2.ptr, x.ptr, y.ptr + 4, t1.ptr) == L.RN_ERR_INVALID      # misaligned output
    assert call(t2.ptr, x.ptr, y.ptr, t1.ptr, w3p=w3.ptr + 2) == L.RN_ERR_INVALID
    assert call(t2.ptr, x.ptr, y.ptr, t1.ptr, scp=sc.ptr + 4) == L.RN_ERR_INVALID
    assert call(t2.ptr, x.ptr, x.ptr, t1.ptr) == L.RN_ERR_INVALID          # y on top of the residual
    assert call(t2.ptr, x.ptr, y.ptr, t2.ptr) == L.RN_ERR_INVALID          # t1 on top of t2
    assert call(t2.ptr, x.ptr, y.ptr, y.ptr) == L.RN_ERR_INVALID           # t1 on top of y
    assert b"alias" in lib.rn_last_error(ctx.handle)


def test_bf16_model_agrees_with_fp32_model(state50, finch, golden_dir):
    m32 = R.NativeModel("resnet50", state=state50)
    m16 = R.NativeModel("resnet50", state=state50, dtype="bf16")
    try:
        x = R.weights.generate_input(6, seed=41)
        x[2] = finch[0]
        a, b = m32.forward(x, fused=True), m16.forward(x, fused=True)
        golden = np.load(os.path.join(golden_dir, "resnet50_finch_logits.npy"))
        # bf16 has 8 significant bits; through 53 layers the logits (|.| <= 3.6) move by ~1e-2
        assert np.abs(a - b).max() <= 0.12
        assert np.abs(b[2:3] - golden).max() <= 0.12
        assert np.array_equal(a.argmax(1), b.argmax(1))  # top-1 agreement (config 4's parity bar)
        # batch invariance and determinism hold in bf16 too
        assert np.array_equal(b[2:3], m16.forward(finch, fused=True))
        assert np.array_equal(b, m16.forward(x, fused=True))
        with pytest.raises(R.RnError):
            m16.forward(x, fused=False)  # bf16 storage exists only with fused epilogues
        assert m16.activation_bytes() * 2 == m32.activation_bytes()
    finally:
        m32.close()
        m16.close()


def test_bf16_fc_chunked_k_sum_is_the_same_whole_or_in_pieces():
    """bf16 operands with an fp32 result (the fc of a bf16 model, [B,2048] x [1000,2048]^T + bias): K = 32
    tiles is summed as eight chunks, ((c0 + c1) + c2) + ..., like the fp32 layers with K >= 1024.  Tiles
    computed whole fold the chunks in registers, a launch that cannot fill the chip cuts every tile into
    (tile, chunk) pieces: the same bits for a row whatever the batch it is in, and the oracle's values on
    the rounded operands."""
    K, N = 2048, 1000
    w = rnd((N, K, 1, 1), 901) / np.sqrt(K)
    bias = rnd((N,), 902)
    big = rnd((700, K, 1, 1), 903)                 # 11 x 16 tiles: a few whole rounds would need B > 1024;
    got = ops.conv2d_nhwc_bf16(big, w, 1, 0, None, bias, None, False, out_f32=True)
    for lo, hi in ((0, 1), (5, 9), (300, 556), (699, 700)):
        part = ops.conv2d_nhwc_bf16(big[lo:hi], w, 1, 0, None, bias, None, False, out_f32=True)
        assert np.array_equal(part, got[lo:hi]), (lo, hi)
    huge = np.concatenate([big] * 6)[:4100]         # 65 x 16 = 1040 tiles: whole tiles and a cut tail
    g2 = ops.conv2d_nhwc_bf16(huge, w, 1, 0, None, bias, None, False, out_f32=True)
    assert np.array_equal(g2[:700], got) and np.array_equal(g2[3500:4100], got[:600])
    ref = O.conv2d(ops.bf16_round(big[:64]), ops.bf16_round(w), 1, 0) + bias[None, :, None, None]
    assert np.abs(got[:64] - ref).max() <= 2e-6 * np.sqrt(K) * float(np.abs(ref).max()) + 1e-6

