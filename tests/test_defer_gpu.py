"""Deferred execution of the reference's op-by-op call sequence (rn_ctx_set_deferred, rn_defer.hip):
the seven reference entry points record their calls, a convolution runs together with the in-place
batch-norm / add / ReLU behind it as ONE fused NHWC launch into the caller's own buffer, and a buffer
gets its NCHW content back when it is observed.  Held against the literal route (one launch per call,
the parity baseline), the oracle and the reference module's golden logits."""
import os
import subprocess

import numpy as np
import pytest

import resnet_c_amd as R
from oracle import oracle as O
from resnet_c_amd import _lib as L

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape, dtype=np.float32) * scale).astype(np.float32)


@pytest.fixture()
def dctx():
    ctx = R.get_ctx()
    ctx.set_deferred(True)
    yield ctx
    ctx.set_deferred(False)
    assert ctx.deferred_stats()["pending_ops"] == 0 and ctx.deferred_stats()["nhwc_buffers"] == 0


def gpu(a):
    return R.FloatTensor.from_numpy(np.ascontiguousarray(a, dtype=np.float32), R.Device.GPU)


def bn_params(C, seed):
    g = np.random.default_rng(seed)
    return (g.random(C, dtype=np.float32) + 0.5, g.standard_normal(C, dtype=np.float32) * 0.1,
            g.standard_normal(C, dtype=np.float32) * 0.1, g.random(C, dtype=np.float32) + 0.5)


def call(ctx, name, *args):
    L.check(getattr(L.lib(), name)(ctx.handle, *args), name, ctx.handle)


def test_whole_network_deferred_matches_goldens_and_the_literal_route(state50, finch, golden_dir, dctx):
    """createResnet / resnetForward (the reference driver object for object, one C-ABI call per reference
    op on NCHW tensors) on a deferred context: 50 fused launches for the 53 convolutions (conv3 of each of layer1's
    blocks runs in one launch with conv1 of the block behind it, the stem convolution with the max-pool), no
    literal batch-norm / ReLU / add pass, no layout pass at all; logits within the bar of the reference module's goldens and within the
    fused epilogue's distance of the literal route; intermediate tensors come back as NCHW when observed."""
    x = np.concatenate([finch, R.weights.generate_input(2, seed=5)])
    dctx.set_deferred(False)
    m = R.createResnet("resnet50", state50)
    xd = gpu(x)
    literal = R.resnetForward(m, xd).numpy().copy()
    lit_l1 = m.layer1.blocks[0].act3_out.numpy().copy()
    lit_t = m.layer3.blocks[2].act2_out.numpy().copy()
    dctx.set_deferred(True)
    s0 = dctx.deferred_stats()
    out = R.resnetForward(m, xd)
    assert dctx.deferred_stats()["pending_ops"] == 174          # nothing has run yet (main.cu: 174 ops)
    got = out.numpy()                                           # observed: the list runs
    s1 = dctx.deferred_stats()
    assert s1["pending_ops"] == 0
    # every convolution with its in-place ops folded in; layer1: conv3 of a block + conv1 of the next, one launch
    assert s1["fused_launches"] - s0["fused_launches"] == 50
    assert s1["literal_launches"] - s0["literal_launches"] == 2  # avg-pool, fc (the max-pool: inside the stem launch)
    assert s1["transposes"] - s0["transposes"] == 0             # the stem launch reads the NCHW image itself
    want = np.load(os.path.join(golden_dir, "resnet50_finch_logits.npy"))
    assert np.abs(got[:1] - want).max() <= 1e-4
    assert np.abs(got - literal).max() <= 5e-5
    assert np.array_equal(got.argmax(1), literal.argmax(1)) and int(got[0].argmax()) == 112
    # tensors inside the network are NHWC in their buffers now; read, they are the literal route's NCHW values
    a = m.layer1.blocks[0].act3_out.numpy()
    assert np.abs(a - lit_l1).max() <= 2e-5 * max(1.0, float(np.abs(lit_l1).max()))
    t = m.layer3.blocks[2].act2_out.numpy()
    assert np.abs(t - lit_t).max() <= 2e-5 * max(1.0, float(np.abs(lit_t).max()))
    assert dctx.deferred_stats()["nhwc_buffers"] > 40           # whole-tensor reads leave the buffers NHWC
    again = R.resnetForward(m, xd).numpy()                      # and the next forward runs on them as before
    assert np.array_equal(again, got)


@pytest.mark.parametrize("arch,convs", [("resnet101", 104), ("resnet152", 155)])
def test_deeper_networks_deferred_match_the_reference_modules_goldens(arch, convs, finch, golden_dir, dctx):
    """ResNet-101 / 152 through the same reference-shaped driver on a deferred context: the logits of the finch
    image and of the two random images within 1e-4 of the reference module's (tests/golden/make_golden.py), the
    same values as the literal route within the folded batch-norm's distance; every convolution in a fused launch
    (three of them shared: layer1's block boundaries)."""
    state = R.weights.generate_state(arch, seed=0)
    x = np.concatenate([finch, R.weights.generate_input(2, seed=7)])
    m = R.createResnet(arch, state)
    xd = gpu(x)
    s0 = dctx.deferred_stats()
    got = R.resnetForward(m, xd).numpy()
    s1 = dctx.deferred_stats()
    assert s1["fused_launches"] - s0["fused_launches"] == convs - 3 and s1["transposes"] - s0["transposes"] == 0
    want = np.load(os.path.join(golden_dir, f"{arch}_finch_logits.npy"))
    want2 = np.load(os.path.join(golden_dir, f"{arch}_rand2_logits.npy"))
    assert np.abs(got[:1] - want).max() <= 1e-4 and np.abs(got[1:] - want2).max() <= 1e-4
    dctx.set_deferred(False)
    literal = R.resnetForward(m, xd).numpy()
    dctx.set_deferred(True)
    assert np.abs(got - literal).max() <= 5e-5 and np.array_equal(got.argmax(1), literal.argmax(1))


@pytest.mark.parametrize("case", [(2, 64, 64, 14, 14, 3, 1, 1), (3, 32, 96, 9, 7, 1, 1, 0), (2, 64, 128, 12, 12, 1, 2, 0),
                                  (1, 128, 32, 7, 7, 3, 2, 1), (2, 3, 64, 32, 32, 7, 2, 3)])
def test_fused_chain_against_the_oracle(case, dctx):
    """conv -> bn (in place) -> add (in place) -> relu (in place) recorded, run as one launch: against the
    oracle's four ops; the residual once as a plain NCHW tensor (transposed on the way in) and once as the
    NHWC output of an earlier deferred convolution."""
    B, Cin, Cout, H, W, k, s, p = case
    seed = sum(case)
    x, w = rnd((B, Cin, H, W), seed), rnd((Cout, Cin, k, k), seed + 1, 1.0 / np.sqrt(Cin * k * k))
    bw, bb, bm, bv = bn_params(Cout, seed + 2)
    ho, wo = O.conv_output_size(H, k, s, p), O.conv_output_size(W, k, s, p)
    res = rnd((B, Cout, ho, wo), seed + 3)
    want = O.relu_(O.add_(O.batchnorm2d_(O.conv2d(x, w, s, p), bw, bb, bm, bv), res))
    dx, dw, dres = gpu(x), gpu(w), gpu(res)
    P = [gpu(v) for v in (bw, bb, bm, bv)]
    out = R.FloatTensor((B, Cout, ho, wo), R.Device.GPU)
    n = B * Cout * ho * wo

    def chain(residual):
        call(dctx, "rn_conv2d_forward", dx.data(), out.data(), dw.data(), k, s, p, ho, wo, B, Cin, Cout, H, W)
        call(dctx, "rn_batchnorm2d_forward", out.data(), out.data(), *(t.data() for t in P), B, Cout, ho * wo)
        call(dctx, "rn_add_forward", out.data(), residual.data(), out.data(), n)
        call(dctx, "rn_relu_forward", out.data(), out.data(), n)

    s0 = dctx.deferred_stats()
    chain(dres)
    got = out.numpy()
    s1 = dctx.deferred_stats()
    assert s1["fused_launches"] - s0["fused_launches"] == 1 and s1["literal_launches"] == s0["literal_launches"]
    tol = 3e-6 * np.sqrt(Cin * k * k) * float(np.abs(want).max()) + 1e-5
    assert np.abs(got - want).max() <= tol
    # the residual produced by a deferred 1x1 convolution of the same shape: NHWC-tagged, used in place
    w2 = rnd((Cout, Cout, 1, 1), seed + 4, 1.0 / np.sqrt(Cout))
    r2 = R.FloatTensor((B, Cout, ho, wo), R.Device.GPU)
    dw2 = gpu(w2)
    if Cout % 32 == 0:
        call(dctx, "rn_conv2d_forward", dres.data(), r2.data(), dw2.data(), 1, 1, 0, ho, wo, B, Cout, Cout, ho, wo)
        t0 = dctx.deferred_stats()["transposes"]
        chain(r2)
        got2 = out.numpy()
        res2 = O.conv2d(res, w2, 1, 0)
        want2 = O.relu_(O.add_(O.batchnorm2d_(O.conv2d(x, w, s, p), bw, bb, bm, bv), res2))
        assert np.abs(got2 - want2).max() <= tol + 3e-6 * np.sqrt(Cout) * float(np.abs(res2).max())
        # layout passes: dres -> NHWC for the 1x1, x -> NHWC for the chain, out -> NCHW for the read; none for r2
        assert dctx.deferred_stats()["transposes"] - t0 == (3 if H * W > 1 else 1)


@pytest.mark.parametrize("next_mid", [64, 128])
def test_conv3_of_a_block_and_conv1_of_the_next_run_as_one_launch(next_mid, dctx):
    """layerForward's block boundary (main.cu:131-164): conv3 -> bn3 -> add -> relu of a 64-channel block, then
    conv1 -> bn1 -> relu of the next block on that output.  Recorded on a deferred context with NHWC-tagged inputs,
    the seven calls run as ONE launch (rn_conv_chain_forward_dt) that writes both tensors; both against the oracle,
    and bit for bit what the two fused launches give (the same calls with an op in between that breaks the
    pattern: a ReLU of an unrelated buffer)."""
    B, H, W = 3, 10, 9
    g = 900 + next_mid
    x0, x1 = rnd((B, 64, H, W), g), rnd((B, 64, H, W), g + 1)
    wa, wb = rnd((64, 64, 1, 1), g + 2, 0.125), rnd((256, 64, 1, 1), g + 3, 0.125)
    w3, w1 = rnd((256, 64, 1, 1), g + 4, 0.125), rnd((next_mid, 256, 1, 1), g + 5, 1 / 16)
    p3, p1 = bn_params(256, g + 6), bn_params(next_mid, g + 7)
    t2_ = O.conv2d(x0, wa, 1, 0)                      # both inputs of the chain come out of deferred convolutions:
    r_ = O.conv2d(x1, wb, 1, 0)                       # NHWC in their buffers when the chain runs
    y_ = O.relu_(O.add_(O.batchnorm2d_(O.conv2d(t2_, w3, 1, 0), *p3), r_))
    t1_ = O.relu_(O.batchnorm2d_(O.conv2d(y_, w1, 1, 0), *p1))
    D = {k: gpu(v) for k, v in dict(x0=x0, x1=x1, wa=wa, wb=wb, w3=w3, w1=w1).items()}
    P3, P1 = [gpu(v) for v in p3], [gpu(v) for v in p1]
    t2, r = R.FloatTensor((B, 64, H, W), R.Device.GPU), R.FloatTensor((B, 256, H, W), R.Device.GPU)
    y, t1 = R.FloatTensor((B, 256, H, W), R.Device.GPU), R.FloatTensor((B, next_mid, H, W), R.Device.GPU)
    other = gpu(rnd((8,), g + 8))
    hw = H * W

    wd = gpu(rnd((32, 256, 1, 1), g + 9, 1 / 16))
    pd = bn_params(32, g + 10)
    Pd = [gpu(v) for v in pd]
    Hd, Wd = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    d = R.FloatTensor((B, 32, Hd, Wd), R.Device.GPU)
    d_ = O.batchnorm2d_(O.conv2d(y_, rnd((32, 256, 1, 1), g + 9, 1 / 16), 2, 0), *pd)

    def program(split, shortcut=False):
        call(dctx, "rn_conv2d_forward", D["x0"].data(), t2.data(), D["wa"].data(), 1, 1, 0, H, W, B, 64, 64, H, W)
        call(dctx, "rn_conv2d_forward", D["x1"].data(), r.data(), D["wb"].data(), 1, 1, 0, H, W, B, 64, 256, H, W)
        call(dctx, "rn_conv2d_forward", t2.data(), y.data(), D["w3"].data(), 1, 1, 0, H, W, B, 64, 256, H, W)
        call(dctx, "rn_batchnorm2d_forward", y.data(), y.data(), *(t.data() for t in P3), B, 256, hw)
        call(dctx, "rn_add_forward", y.data(), r.data(), y.data(), B * 256 * hw)
        call(dctx, "rn_relu_forward", y.data(), y.data(), B * 256 * hw)
        if split:
            call(dctx, "rn_relu_forward", other.data(), other.data(), 8)
        if shortcut:   # the next stage's projection shortcut (1x1 / 2 + bn) stands between, as main.cu:131-137 runs it
            call(dctx, "rn_conv2d_forward", y.data(), d.data(), wd.data(), 1, 2, 0, Hd, Wd, B, 256, 32, H, W)
            call(dctx, "rn_batchnorm2d_forward", d.data(), d.data(), *(t.data() for t in Pd), B, 32, Hd * Wd)
        call(dctx, "rn_conv2d_forward", y.data(), t1.data(), D["w1"].data(), 1, 1, 0, H, W, B, 256, next_mid, H, W)
        call(dctx, "rn_batchnorm2d_forward", t1.data(), t1.data(), *(t.data() for t in P1), B, next_mid, hw)
        call(dctx, "rn_relu_forward", t1.data(), t1.data(), B * next_mid * hw)
        s0 = dctx.deferred_stats()
        dctx.flush()
        s1 = dctx.deferred_stats()
        return s1["fused_launches"] - s0["fused_launches"], y.numpy().copy(), t1.numpy().copy()

    n_two, y_two, t1_two = program(True)
    n_one, y_one, t1_one = program(False)
    assert (n_two, n_one) == (4, 3)
    assert np.array_equal(y_one, y_two) and np.array_equal(t1_one, t1_two)
    n_sc, y_sc, t1_sc = program(False, shortcut=True)           # chain launch, then the shortcut: 4 launches, not 5
    assert n_sc == 4 and np.array_equal(y_sc, y_one) and np.array_equal(t1_sc, t1_one)
    assert np.abs(d.numpy() - d_).max() <= 3e-6 * 16 * float(np.abs(d_).max()) + 2e-5
    assert np.abs(y_one - y_) .max() <= 3e-6 * 8 * float(np.abs(y_).max()) + 1e-5
    assert np.abs(t1_one - t1_).max() <= 3e-6 * 16 * float(np.abs(t1_).max()) + 2e-5
    # a residual that is a plain NCHW tensor (not produced on this context): the two-launch form, same values
    r_plain = gpu(r_)
    call(dctx, "rn_conv2d_forward", t2.data(), y.data(), D["w3"].data(), 1, 1, 0, H, W, B, 64, 256, H, W)
    call(dctx, "rn_batchnorm2d_forward", y.data(), y.data(), *(t.data() for t in P3), B, 256, hw)
    call(dctx, "rn_add_forward", y.data(), r_plain.data(), y.data(), B * 256 * hw)
    call(dctx, "rn_relu_forward", y.data(), y.data(), B * 256 * hw)
    call(dctx, "rn_conv2d_forward", y.data(), t1.data(), D["w1"].data(), 1, 1, 0, H, W, B, 256, next_mid, H, W)
    call(dctx, "rn_batchnorm2d_forward", t1.data(), t1.data(), *(t.data() for t in P1), B, next_mid, hw)
    call(dctx, "rn_relu_forward", t1.data(), t1.data(), B * next_mid * hw)
    s0 = dctx.deferred_stats()
    dctx.flush()
    assert dctx.deferred_stats()["fused_launches"] - s0["fused_launches"] == 2
    assert np.abs(t1.numpy() - t1_).max() <= 3e-6 * 16 * float(np.abs(t1_).max()) + 2e-5


@pytest.mark.parametrize("case", [(2, 3, 32, 32, True), (1, 3, 23, 48, True), (3, 1, 9, 16, True), (2, 3, 16, 20, False),
                                  (2, 3, 32, 32, "no relu"), (2, 3, 32, 32, "reads between")])
def test_stem_and_its_max_pool_run_as_one_launch_that_writes_both(case, dctx):
    """The reference's first four ops (main.cu:179-192): conv 7x7 / 2 / 3 -> bn -> relu in place on `act1`, max-pool
    3x3 / 2 / 1 of it into another tensor.  Deferred, they are ONE launch (rn_stem_conv_pool_nchw_forward) that reads
    the NCHW image itself and writes BOTH tensors -- the stem tensor is the caller's and must hold its value -- when
    the image suits the fused stem (W % 4 == 0, conv output width a multiple of 8); otherwise, or without the ReLU
    (the pool is an integer maximum of non-negative bit patterns), or with a read in between, the separate launches.
    Both tensors against the oracle; the pooled tensor is exactly the oracle's max-pool of the stem tensor read back."""
    B, Cin, H, W, mode = case
    x, w = rnd((B, Cin, H, W), 700 + H), rnd((64, Cin, 7, 7), 701 + H, 1.0 / np.sqrt(49 * Cin))
    prm = bn_params(64, 702 + H)
    ho, wo = O.conv_output_size(H, 7, 2, 3), O.conv_output_size(W, 7, 2, 3)
    ph, pw = O.conv_output_size(ho, 3, 2, 1), O.conv_output_size(wo, 3, 2, 1)
    y_ = O.batchnorm2d_(O.conv2d(x, w, 2, 3), *prm)
    if mode != "no relu":
        y_ = O.relu_(y_)
    p_ = O.maxpool2d(y_, 3, 2, 1)
    dx, dw, P = gpu(x), gpu(w), [gpu(v) for v in prm]
    y, pl = R.FloatTensor((B, 64, ho, wo), R.Device.GPU), R.FloatTensor((B, 64, ph, pw), R.Device.GPU)
    for _ in range(2):   # the second pass finds both tensors NHWC-tagged from the first
        call(dctx, "rn_conv2d_forward", dx.data(), y.data(), dw.data(), 7, 2, 3, ho, wo, B, Cin, 64, H, W)
        call(dctx, "rn_batchnorm2d_forward", y.data(), y.data(), *(t.data() for t in P), B, 64, ho * wo)
        if mode != "no relu":
            call(dctx, "rn_relu_forward", y.data(), y.data(), B * 64 * ho * wo)
        if mode == "reads between":
            assert np.abs(y.numpy() - y_).max() <= 2e-5 * max(1.0, float(np.abs(y_).max()))
        call(dctx, "rn_maxpool2d_forward", y.data(), pl.data(), 3, 2, 1, ph, pw, B, 64, ho, wo)
        s0 = dctx.deferred_stats()
        dctx.flush()
        s1 = dctx.deferred_stats()
        d = {k: s1[k] - s0[k] for k in ("fused_launches", "literal_launches", "transposes")}
        if mode is True:
            assert d == {"fused_launches": 1, "literal_launches": 0, "transposes": 0}
        elif mode == "reads between":
            assert d == {"fused_launches": 0, "literal_launches": 1, "transposes": 0}   # only the pool was pending
        else:
            assert d == {"fused_launches": 1, "literal_launches": 1, "transposes": 1}
        got_p, got_y = pl.numpy(), y.numpy()
        tol = 3e-6 * np.sqrt(49 * Cin) * float(np.abs(y_).max()) + 1e-5
        assert np.abs(got_y - y_).max() <= tol and np.abs(got_p - p_).max() <= tol
        assert np.array_equal(got_p, O.maxpool2d(got_y, 3, 2, 1))


def test_sequences_that_do_not_fold_run_literally_in_call_order(dctx):
    """Only conv -> [bn] -> [add] -> [relu], each in place on the convolution's output, folds.  Another
    order, another buffer, a shape without an NHWC contraction: literal launches in call order, same
    values as the literal route bit for bit (they ARE the literal kernels; on NHWC-tagged inputs the NHWC
    forms of batch-norm / ReLU / add / pools, which are bit-identical to the NCHW forms)."""
    B, C, H, W = 2, 32, 10, 10
    x, w = rnd((B, C, H, W), 1), rnd((C, C, 3, 3), 2, 0.06)
    bw, bb, bm, bv = bn_params(C, 3)
    other = rnd((B, C, H, W), 4)
    w5 = rnd((8, 5, 3, 3), 5, 0.2)
    x5 = rnd((B, 5, H, W), 6)

    def program(ctx):
        dx, dw, dother, dw5, dx5 = gpu(x), gpu(w), gpu(other), gpu(w5), gpu(x5)
        P = [gpu(v) for v in (bw, bb, bm, bv)]
        t = R.FloatTensor((B, C, H, W), R.Device.GPU)
        u = R.FloatTensor((B, C, H, W), R.Device.GPU)
        v = R.FloatTensor((B, 8, H, W), R.Device.GPU)
        pool = R.FloatTensor((B, C, 5, 5), R.Device.GPU)
        n = B * C * H * W
        call(ctx, "rn_conv2d_forward", dx.data(), t.data(), dw.data(), 3, 1, 1, H, W, B, C, C, H, W)
        call(ctx, "rn_relu_forward", t.data(), t.data(), n)                       # relu BEFORE bn: relu folds, bn does not
        call(ctx, "rn_batchnorm2d_forward", t.data(), t.data(), *(q.data() for q in P), B, C, H * W)
        call(ctx, "rn_batchnorm2d_forward", t.data(), u.data(), *(q.data() for q in P), B, C, H * W)   # out of place
        call(ctx, "rn_add_forward", u.data(), dother.data(), u.data(), n)         # NHWC-tagged + plain NCHW operand
        call(ctx, "rn_maxpool2d_forward", t.data(), pool.data(), 3, 2, 1, 5, 5, B, C, H, W)
        call(ctx, "rn_conv2d_forward", dx5.data(), v.data(), dw5.data(), 3, 1, 1, H, W, B, 5, 8, H, W)  # Cin = 5: direct kernel
        call(ctx, "rn_relu_forward", v.data(), v.data(), B * 8 * H * W)
        return [a.numpy() for a in (t, u, pool, v)]

    dctx.set_deferred(False)
    want = program(dctx)
    dctx.set_deferred(True)
    got = program(dctx)
    ref_t = O.batchnorm2d(O.relu(O.conv2d(x, w, 1, 1)), bw, bb, bm, bv)
    assert np.abs(want[0] - ref_t).max() <= 3e-5
    for g, wnt in zip(got, want):
        assert np.array_equal(g, wnt)


def test_observation_rules(dctx):
    """What is read is NCHW, whatever the buffer holds: whole-tensor copies (the buffer stays NHWC), partial
    copies and device-to-device copies (the buffer is rewritten first), rn_observe; writes into a recorded
    operand run the list first; a free with recorded ops pending; rewritten batch-norm parameters and
    weights are folded / packed again."""
    B, C, H, W = 2, 64, 6, 6
    x, w = rnd((B, C, H, W), 11), rnd((C, C, 1, 1), 12, 0.12)
    bw, bb, bm, bv = bn_params(C, 13)
    dx, dw = gpu(x), gpu(w)
    P = [gpu(v) for v in (bw, bb, bm, bv)]
    out = R.FloatTensor((B, C, H, W), R.Device.GPU)
    n = B * C * H * W

    def run():
        call(dctx, "rn_conv2d_forward", dx.data(), out.data(), dw.data(), 1, 1, 0, H, W, B, C, C, H, W)
        call(dctx, "rn_batchnorm2d_forward", out.data(), out.data(), *(q.data() for q in P), B, C, H * W)
        call(dctx, "rn_relu_forward", out.data(), out.data(), n)

    want = O.relu_(O.batchnorm2d_(O.conv2d(x, w, 1, 0), bw, bb, bm, bv))
    run()
    whole = out.numpy()
    assert np.abs(whole - want).max() <= 1e-5 and dctx.deferred_stats()["nhwc_buffers"] >= 1
    # a partial read: the second image only
    host = np.empty(C * H * W, dtype=np.float32)
    call(dctx, "rn_memcpy_d2h", host.ctypes.data, out.data() + C * H * W * 4, C * H * W * 4)
    assert np.array_equal(host.reshape(C, H, W), whole[1])
    assert np.array_equal(out.numpy(), whole)
    # device-to-device copy of a tagged buffer
    run()
    twin = R.FloatTensor((B, C, H, W), R.Device.GPU)
    call(dctx, "rn_memcpy_d2d", twin.data(), out.data(), n * 4)
    assert np.array_equal(twin.numpy(), whole)
    # rn_observe: the caller's own kernels may read the pointer afterwards (here: a raw hipMemcpy via torch-free d2h)
    run()
    dctx.observe(out.data())
    assert dctx.deferred_stats()["pending_ops"] == 0
    t0 = dctx.deferred_stats()["transposes"]
    assert np.array_equal(out.numpy(), whole) and dctx.deferred_stats()["transposes"] == t0
    # a write into the input while ops that read it are recorded: they run first, on the old content
    run()
    x2 = rnd((B, C, H, W), 14)
    call(dctx, "rn_memcpy_h2d", dx.data(), x2.ctypes.data, n * 4)
    assert np.array_equal(out.numpy(), whole)
    run()
    want2 = O.relu_(O.batchnorm2d_(O.conv2d(x2, w, 1, 0), bw, bb, bm, bv))
    assert np.abs(out.numpy() - want2).max() <= 1e-5
    # new batch-norm parameters in the same buffers: folded again, not taken from the cache
    bw2 = (bw * 2).astype(np.float32)
    call(dctx, "rn_memcpy_h2d", P[0].data(), bw2.ctypes.data, C * 4)
    run()
    want3 = O.relu_(O.batchnorm2d_(O.conv2d(x2, w, 1, 0), bw2, bb, bm, bv))
    assert np.abs(out.numpy() - want3).max() <= 1e-5
    # new weights in the same buffer: packed again
    w2 = rnd((C, C, 1, 1), 15, 0.12)
    call(dctx, "rn_memcpy_h2d", dw.data(), w2.ctypes.data, w2.nbytes)
    run()
    want4 = O.relu_(O.batchnorm2d_(O.conv2d(x2, w2, 1, 0), bw2, bb, bm, bv))
    assert np.abs(out.numpy() - want4).max() <= 1e-5
    # memset over a tagged buffer, then a read: zeros, and the tag is gone
    run()
    dctx.flush()
    call(dctx, "rn_memset", out.data(), 0, n * 4)
    assert not out.numpy().any()
    # a free while ops that name the buffer are recorded
    tmp = R.FloatTensor((B, C, H, W), R.Device.GPU)
    call(dctx, "rn_conv2d_forward", dx.data(), tmp.data(), dw.data(), 1, 1, 0, H, W, B, C, C, H, W)
    call(dctx, "rn_relu_forward", tmp.data(), out.data(), n)
    del tmp
    assert dctx.deferred_stats()["pending_ops"] == 0
    assert np.abs(out.numpy() - O.relu(O.conv2d(x2, w2, 1, 0))).max() <= 1e-5
    # entry points outside the seven see NCHW: the layout converter on a tagged buffer
    run()
    dctx.flush()
    nhwc = R.FloatTensor((B, H, W, C), R.Device.GPU)
    call(dctx, "rn_nchw_to_nhwc", out.data(), nhwc.data(), B, C, H, W)
    assert np.abs(nhwc.numpy().transpose(0, 3, 1, 2) - want4).max() <= 1e-5


def test_operands_that_are_slices_of_an_nhwc_tagged_buffer(dctx):
    """A caller may hand a later op a PART of a tensor an earlier (fused, NHWC-writing) convolution produced: the
    second image of the batch as a convolution input or a residual, an in-place ReLU on the first image only, an
    output that covers part of a tagged buffer.  In the caller's NCHW arithmetic those are plain pointer offsets:
    the tagged buffer gets its NCHW content back before such an op runs.  Against the literal route, bit for bit
    (no batch-norm is folded here)."""
    B, C, H, W = 3, 32, 8, 8
    x, w = rnd((B, C, H, W), 21), rnd((C, C, 3, 3), 22, 0.06)
    w1 = rnd((64, C, 1, 1), 23, 0.17)
    img = C * H * W

    def program(ctx):
        dx, dw, dw1 = gpu(x), gpu(w), gpu(w1)
        a = R.FloatTensor((B, C, H, W), R.Device.GPU)
        b = R.FloatTensor((1, 64, H, W), R.Device.GPU)
        c = R.FloatTensor((B, C, H, W), R.Device.GPU)
        call(ctx, "rn_memset", c.data(), 0, B * img * 4)
        call(ctx, "rn_conv2d_forward", dx.data(), a.data(), dw.data(), 3, 1, 1, H, W, B, C, C, H, W)       # a: NHWC-tagged
        call(ctx, "rn_conv2d_forward", a.data() + img * 4, b.data(), dw1.data(), 1, 1, 0, H, W, 1, C, 64, H, W)  # image 1 of a
        call(ctx, "rn_conv2d_forward", dx.data(), a.data(), dw.data(), 3, 1, 1, H, W, B, C, C, H, W)       # tagged again
        call(ctx, "rn_relu_forward", a.data(), a.data(), img)                                               # image 0 only
        call(ctx, "rn_conv2d_forward", dx.data(), a.data(), dw.data(), 3, 1, 1, H, W, B, C, C, H, W)
        call(ctx, "rn_add_forward", c.data() + 2 * img * 4, a.data() + img * 4, c.data() + 2 * img * 4, img)  # slices of both
        call(ctx, "rn_conv2d_forward", dx.data(), c.data(), dw.data(), 3, 1, 1, H, W, 2, C, C, H, W)       # covers 2 of c's 3 images
        return [t.numpy() for t in (a, b, c)]

    dctx.set_deferred(False)
    want = program(dctx)
    dctx.set_deferred(True)
    got = program(dctx)
    for g_, w_ in zip(got, want):
        assert np.array_equal(g_, w_)


def test_operands_tagged_under_another_shape(dctx):
    """The same device buffer seen under two shapes (Tensor::view shares storage): a convolution output
    [B, 64, 8, 8] -- NHWC-tagged -- is handed on as [B, 32, 8, 16], as the INPUT of the next convolution and as
    the RESIDUAL of another.  NCHW arithmetic on the caller's side: the buffer goes back to NCHW before either
    use, and both rewrites happen before the launch's own scratch (the transposed residual) is in use."""
    B = 2
    x, w = rnd((B, 32, 8, 8), 31), rnd((64, 32, 3, 3), 32, 0.06)
    w2 = rnd((32, 32, 1, 1), 33, 0.17)
    other = rnd((B, 32, 8, 16), 34)

    def program(ctx):
        dx, dw, dw2, dother = gpu(x), gpu(w), gpu(w2), gpu(other)
        a = R.FloatTensor((B, 64, 8, 8), R.Device.GPU)        # also read as [B, 32, 8, 16]
        r = R.FloatTensor((B, 64, 8, 8), R.Device.GPU)        # likewise: the residual
        y = R.FloatTensor((B, 32, 8, 16), R.Device.GPU)
        n = B * 32 * 8 * 16
        call(ctx, "rn_conv2d_forward", dx.data(), a.data(), dw.data(), 3, 1, 1, 8, 8, B, 32, 64, 8, 8)    # a tagged [B,64,8,8]
        call(ctx, "rn_conv2d_forward", dx.data(), r.data(), dw.data(), 3, 1, 1, 8, 8, B, 32, 64, 8, 8)    # r tagged [B,64,8,8]
        call(ctx, "rn_relu_forward", r.data(), r.data(), n)
        call(ctx, "rn_conv2d_forward", a.data(), y.data(), dw2.data(), 1, 1, 0, 8, 16, B, 32, 32, 8, 16)  # a as [B,32,8,16]
        call(ctx, "rn_add_forward", y.data(), r.data(), y.data(), n)                                      # r as [B,32,8,16]
        call(ctx, "rn_relu_forward", y.data(), y.data(), n)
        return [t.numpy() for t in (a, r)] + [y.numpy()]

    dctx.set_deferred(False)
    want = program(dctx)
    dctx.set_deferred(True)
    got = program(dctx)
    for g_, w_ in zip(got, want):
        assert np.array_equal(g_, w_)


def _build(tmp_path, name):
    exe = str(tmp_path / name)
    libdir = os.path.dirname(R._lib.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", f"-I{ROOT}/include", f"{ROOT}/examples/{name}.cpp",
                    f"-L{libdir}", "-lrn_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                   check=True, capture_output=True, text=True)
    return exe


def test_cpp_veneer_is_deferred_by_default_and_literal_on_request(tmp_path, state50, finch, golden_dir):
    """examples/resnet_veneer.cpp (reference-named C++ classes, tensors allocated and freed per block): the
    veneer's context is deferred unless RN_VENEER_LITERAL=1.  Both within the bar of the goldens, the same
    class indices; the literal run is the op-by-op mode of the model driver to 1e-5."""
    exe = _build(tmp_path, "resnet_veneer")
    os.mkdir(tmp_path / "weights_bin")
    R.weights.save_weights_bin(state50, str(tmp_path / "weights_bin"))
    x = np.concatenate([finch, R.weights.generate_input(1, seed=7)[:1]]).astype(np.float32)
    x.tofile(tmp_path / "input.bin")
    outs = {}
    for mode, env in (("deferred", {}), ("literal", {"RN_VENEER_LITERAL": "1"})):
        r = subprocess.run([exe, "50", "input.bin", f"logits_{mode}.bin"], cwd=tmp_path, capture_output=True,
                           text=True, timeout=300, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr
        outs[mode] = np.fromfile(tmp_path / f"logits_{mode}.bin", dtype=np.float32).reshape(2, 1000)
        lines = [l for l in r.stdout.splitlines() if l.startswith("max index is")]
        assert lines == [f"max index is {int(outs[mode][0].argmax())}", f"max index is {int(outs[mode][1].argmax())}"]
    want = np.load(os.path.join(golden_dir, "resnet50_finch_logits.npy"))
    for mode in outs:
        assert np.abs(outs[mode][:1] - want).max() <= 1e-4 and int(outs[mode][0].argmax()) == 112
    assert np.abs(outs["deferred"] - outs["literal"]).max() <= 5e-5
    assert np.array_equal(outs["deferred"].argmax(1), outs["literal"].argmax(1))
    m = R.NativeModel("resnet50", state=state50)
    try:
        assert np.abs(m.forward(x, fused=False) - outs["literal"]).max() <= 1e-5
    finally:
        m.close()
