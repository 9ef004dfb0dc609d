"""Whole-network parity on the GPU (configs 2, 3 and 5 of BASELINE.md).

The north star's bar: logits within 1e-4 (fp32) of the reference's PyTorch module
on the same exported weights and test image, top-1 identical.  Golden logits come
from the reference module itself (tests/golden/make_golden.py).
"""
import json
import os
import subprocess

import numpy as np
import pytest

import resnet_c_amd as R
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def model50(state50):
    m = R.NativeModel("resnet50", state=state50)
    yield m
    m.close()


@pytest.mark.parametrize("fused", [False, True])
def test_resnet50_b1_logits_vs_reference_golden(model50, finch, golden_dir, fused):
    got = model50.forward(finch, fused=fused)
    want = np.load(os.path.join(golden_dir, "resnet50_finch_logits.npy"))
    want64 = np.load(os.path.join(golden_dir, "resnet50_finch_logits_f64.npy"))
    assert got.shape == (1, 1000)
    assert np.abs(got - want).max() <= TOL
    assert np.abs(got - want64).max() <= TOL
    assert R.ops.argmax(got)[0] == int(want.argmax(1)[0]) == 112


@pytest.mark.parametrize("fused", [False, True])
def test_resnet152_b1_logits_vs_reference_golden(state152, finch, golden_dir, fused):
    m = R.NativeModel("resnet152", state=state152)
    try:
        got = m.forward(finch, fused=fused)
    finally:
        m.close()
    want = np.load(os.path.join(golden_dir, "resnet152_finch_logits.npy"))
    info = json.load(open(os.path.join(golden_dir, "resnet152_taps.json")))
    assert np.abs(got - want).max() <= TOL
    assert R.ops.argmax(got)[0] == info["finch_top1"]


def test_resnet50_random_pair_vs_golden_and_oracle(model50, state50, golden_dir):
    x = R.weights.generate_input(2, seed=7)
    want = np.load(os.path.join(golden_dir, "resnet50_rand2_logits.npy"))
    for fused in (False, True):
        got = model50.forward(x, fused=fused)
        assert np.abs(got - want).max() <= TOL
        assert np.array_equal(R.ops.argmax(got), want.argmax(1))
    cpu = O.resnet_forward(state50, x, "resnet50")
    assert np.abs(model50.forward(x, fused=False) - cpu).max() <= TOL


def test_batch_invariance_is_bit_exact(model50, finch):
    """No op reduces across the batch: row i of a batch equals the
    batch-1 result of image i bit for bit, whatever the tile the row lands in."""
    x = R.weights.generate_input(5, seed=11)
    x[3] = finch[0]
    for fused in (False, True):
        full = model50.forward(x, fused=fused)
        for i in (0, 3, 4):
            assert np.array_equal(full[i:i + 1], model50.forward(x[i:i + 1], fused=fused))
        assert np.array_equal(full, model50.forward(x, fused=fused))  # deterministic


def test_fused_and_reference_op_sequence_agree(model50):
    x = R.weights.generate_input(3, seed=21)
    a, b = model50.forward(x, fused=False), model50.forward(x, fused=True)
    assert np.abs(a - b).max() <= 2e-5
    assert np.array_equal(a.argmax(1), b.argmax(1))


def test_tile_tuning_changes_speed_not_results(model50):
    """Every tile candidate of the contraction sums each output in the same order, so the
    tuned model is bit-identical to the untuned one (and so is every forced candidate)."""
    from resnet_c_amd import _lib as L
    x = R.weights.generate_input(4, seed=31)
    base = model50.forward(x, fused=True)
    xin = R.FloatTensor.from_numpy(x, R.Device.GPU)
    out = R.FloatTensor((4, 1000), R.Device.GPU)
    model50.tune(xin.data(), 4, out.data(), fused=True)
    model50.ctx.sync()
    assert np.array_equal(out.numpy(), base)
    assert np.array_equal(model50.forward(x, fused=True), base)
    w = np.random.default_rng(1).standard_normal((96, 64, 3, 3), dtype=np.float32)
    xs = np.random.default_rng(2).standard_normal((3, 64, 9, 9), dtype=np.float32)
    ref = R.ops.conv2d(xs, w, 1, 1, "nhwc")
    ctx = R.get_ctx()
    for cand in range(1, L.lib().rn_conv_tile_candidates() + 1):
        L.check(L.lib().rn_ctx_set_conv_tile(ctx.handle, cand), "tile", ctx.handle)
        try:
            assert np.array_equal(R.ops.conv2d(xs, w, 1, 1, "nhwc"), ref), cand
        finally:
            L.lib().rn_ctx_set_conv_tile(ctx.handle, 0)
    assert L.lib().rn_ctx_set_conv_tile(ctx.handle, 99) == L.RN_ERR_INVALID


def test_profile_accounts_for_every_reference_op(model50, finch):
    model50.set_profiling(True)
    try:
        model50.forward(finch, fused=False)
        recs = model50.profile()
    finally:
        model50.set_profiling(False)
    count = lambda op: sum(r["op"] == op for r in recs)
    # 53 conv + 53 BN + 49 ReLU + 16 add + maxpool + avgpool + fc = 174
    assert (count("conv2d"), count("batchnorm2d"), count("relu"), count("add")) == (53, 53, 49, 16)
    assert count("maxpool2d") == count("avgpool2d") == count("linear") == 1
    flops = sum(r["flops"] for r in recs)
    assert abs(flops - 8.178368512e9) < 1.0  # BASELINE.md section 2, per image
    by = lambda op: sum(r["bytes"] for r in recs if r["op"] == op)
    assert abs(by("relu") - 76.87e6) < 0.05e6 and abs(by("add") - 66.23e6) < 0.05e6
    assert abs(by("batchnorm2d") - 88.91e6) < 0.6e6
    assert all(r["ms"] >= 0 for r in recs)


def test_reference_shaped_python_graph_nchw(state50, finch, golden_dir):
    """createResnet / resnetForward: the reference driver object for object, one
    C-ABI call per reference op on NCHW tensors (the literal drop-in boundary)."""
    m = R.createResnet("resnet50", state50)
    x = R.FloatTensor.from_numpy(finch, R.Device.GPU)
    out = R.resnetForward(m, x)
    R.get_ctx().sync()
    got = out.numpy()
    want = np.load(os.path.join(golden_dir, "resnet50_finch_logits.npy"))
    assert np.abs(got - want).max() <= TOL
    first_buf = m.layer1.blocks[0].act1_out.data()
    R.resnetForward(m, x)  # second forward allocates nothing (main.cu:141-159)
    assert m.layer1.blocks[0].act1_out.data() == first_buf
    assert m.layer1.blocks[0].downsample is not None and m.layer1.blocks[1].downsample is None
    assert R.model.argmax(got)[0] == 112


def test_plain_c_driver_and_weights_bin_loader(state50, finch, tmp_path):
    """rn_infer is plain C over the C-ABI: weights_bin/ directory in, 'max index is N' out
    (the reference's main(), cuda/inference/main.cu:228-254)."""
    wdir = tmp_path / "weights_bin"
    R.weights.save_weights_bin(state50, str(wdir))
    np.zeros(1, np.float32).tofile(wdir / "bn1.num_batches_tracked")  # ignored, like the reference
    inp = tmp_path / "ILSVRC2012_val_00004749.bin"
    finch.tofile(inp)
    exe = os.path.join(os.path.dirname(R._lib.LIB_PATH), "rn_infer")
    for mode in ("fused", "ops"):
        r = subprocess.run([exe, "--arch", "50", "--weights", str(wdir), "--input", str(inp), "--mode", mode],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert "max index is 112" in r.stdout
    bad = subprocess.run([exe, "--arch", "50", "--weights", str(tmp_path / "nope"), "--input", str(inp)],
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "Can't open" in bad.stderr
    m = R.NativeModel("resnet50", weights_dir=str(wdir))
    try:
        assert int(m.forward(finch).argmax(1)[0]) == 112
    finally:
        m.close()


def test_full_size_batch_properties(model50, finch):
    """Config 3 size (B=256): the oracle would take minutes here, so check
    size-independent properties: planted images reproduce their batch-1 logits bit
    for bit, the batch is deterministic, identical images give identical rows."""
    B = 256
    x = R.weights.generate_input(B, seed=3)
    x[17] = finch[0]
    x[255] = x[0]
    full = model50.forward(x, fused=True)
    assert full.shape == (B, 1000) and np.isfinite(full).all()
    assert np.array_equal(full[17:18], model50.forward(finch, fused=True))
    assert np.array_equal(full[255], full[0])
    assert np.array_equal(full[100:101], model50.forward(x[100:101], fused=True))
    assert np.array_equal(full, model50.forward(x, fused=True))
    assert model50.activation_bytes() < 4 * 2**30


def _full_size_properties(m, finch, B):
    """Size-independent checks at a BASELINE.json batch size: the planted reference image and a
    planted random image reproduce their batch-1 logits bit for bit (batch invariance is what
    makes the sharded run equal to the whole one), duplicates give equal rows, two runs agree."""
    x = R.weights.generate_input(B, seed=3 + B)
    x[17] = finch[0]
    x[B - 1] = x[0]
    full = m.forward(x, fused=True)
    assert full.shape == (B, 1000) and np.isfinite(full).all()
    alone = m.forward(finch, fused=True)
    assert np.array_equal(full[17:18], alone)
    assert np.array_equal(full[B - 1], full[0])
    assert np.array_equal(full[B // 2:B // 2 + 1], m.forward(x[B // 2:B // 2 + 1], fused=True))
    assert np.array_equal(full, m.forward(x, fused=True))
    return full, alone


def test_full_size_resnet152_b128_properties(state152, finch, golden_dir):
    """BASELINE.json configs[4]: ResNet-152 fp32 B=128 -- the cut-tail / chunked-K launch
    geometry of this size, after a tuning pass as bench.py runs it."""
    m = R.NativeModel("resnet152", state=state152)
    try:
        B = 128
        _full_size_properties(m, finch, B)
        x = R.weights.generate_input(B, seed=3 + B)
        x[17] = finch[0]
        x[B - 1] = x[0]
        before = m.forward(x, fused=True)
        xin = R.FloatTensor.from_numpy(x, R.Device.GPU)
        out = R.FloatTensor((B, 1000), R.Device.GPU)
        m.tune(xin.data(), B, out.data(), fused=True)
        m.ctx.sync()
        assert np.array_equal(out.numpy(), before)       # tuned tiles: same bits
        want = np.load(os.path.join(golden_dir, "resnet152_finch_logits.npy"))
        assert np.abs(before[17:18] - want).max() <= TOL   # and the reference module's logits
        assert int(before[17].argmax()) == int(want.argmax(1)[0])
    finally:
        m.close()


def test_full_size_resnet50_bf16_b256_properties(state50, model50, finch):
    """BASELINE.json configs[3], one GPU's shard: ResNet-50 bf16 B=256, tuned as bench.py runs it
    (the 256-wide LDS-DMA tiles are chosen at this size)."""
    m = R.NativeModel("resnet50", state=state50, dtype="bf16")
    try:
        B = 256
        full, alone = _full_size_properties(m, finch, B)
        x = R.weights.generate_input(B, seed=3 + B)
        x[17] = finch[0]
        x[B - 1] = x[0]
        xin = R.FloatTensor.from_numpy(x, R.Device.GPU)
        out = R.FloatTensor((B, 1000), R.Device.GPU)
        m.tune(xin.data(), B, out.data(), fused=True)
        m.ctx.sync()
        assert np.array_equal(out.numpy(), full)
        assert np.array_equal(m.forward(x, fused=True), full)
        # against the fp32 engine: same top-1 wherever the fp32 margin is not razor thin
        f32 = model50.forward(x[:32], fused=True)
        top2 = np.sort(f32, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 0.2
        assert clear.sum() >= 16
        assert np.array_equal(full[:32].argmax(1)[clear], f32.argmax(1)[clear])
        assert np.abs(full[:32] - f32).max() <= 0.25
    finally:
        m.close()


def test_batch_larger_than_one_launch_can_address_runs_as_sub_batches(model50, finch):
    """The reference has no batch limit but memory (main.cu:168-226).  The stem output of 669
    fp32 images passes the kernels' 2^29-element range, so rn_model_forward runs B = 700 as
    sub-batches of at most 512: every row equals its batch-1 bits, and the rate stays the
    engine's."""
    B = 700
    x = np.empty((B, 3, 224, 224), dtype=np.float32)
    base = R.weights.generate_input(100, seed=91)
    for i in range(0, B, 100):
        x[i:i + 100] = base
    x[511] = finch[0]      # last image of the first sub-batch
    x[512] = base[7]       # first image of the second
    x[699] = finch[0]
    xin = R.FloatTensor.from_numpy(x, R.Device.GPU)
    out = R.FloatTensor((B, 1000), R.Device.GPU)
    model50.forward_ptr(xin.data(), B, out.data(), True)
    model50.ctx.sync()
    got = out.numpy()
    assert np.isfinite(got).all()
    alone = model50.forward(finch, fused=True)
    assert np.array_equal(got[511:512], alone) and np.array_equal(got[699:700], alone)
    assert np.array_equal(got[512:513], model50.forward(base[7:8], fused=True))
    assert np.array_equal(got[600:700][:99], got[0:99])   # the same 100 images again
    model50.tune(xin.data(), B, out.data(), True)
    model50.ctx.sync()
    assert np.array_equal(out.numpy(), got)
    del xin, out   # the rate of this batch (14.2k images/s) is tools/latency.py's business, not a parity test's


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_streams_and_depth_first_front_change_speed_not_results(state50, finch, dtype):
    """rn_model_set_streams (a batch as parts on streams of their own) and
    rn_model_set_front_parts (stem + max-pool + first stage in slices, for Infinity-Cache reuse)
    reschedule the same launches on sub-batches: bit-identical logits, also after tuning (the
    tuned tile of a layer is remembered with the launch batch it was tuned at)."""
    m = R.NativeModel("resnet50", state=state50, dtype=dtype)
    try:
        B = 128
        x = R.weights.generate_input(B, seed=66)
        x[77] = finch[0]
        m.set_streams(1)
        base = m.forward(x, fused=True)
        assert np.array_equal(base[77:78], m.forward(finch, fused=True))
        for streams, front in ((2, 1), (1, 4), (2, 4), (4, 2), (1, 8)):
            m.set_streams(streams)
            m.set_front_parts(front)
            assert np.array_equal(m.forward(x, fused=True), base), (streams, front)
        xin = R.FloatTensor.from_numpy(x, R.Device.GPU)
        out = R.FloatTensor((B, 1000), R.Device.GPU)
        m.set_streams(2)
        m.set_front_parts(2)
        m.tune(xin.data(), B, out.data(), fused=True)
        m.ctx.sync()
        assert np.array_equal(out.numpy(), base)
        assert np.array_equal(m.forward(x, fused=True), base)
        with pytest.raises(R.RnError):
            m.set_front_parts(3)
        with pytest.raises(R.RnError):
            m.set_streams(3)
    finally:
        m.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_stem_pool_changes_launches_not_results(state50, finch, golden_dir, dtype):
    """conv1 + bn1 + relu + maxpool as one launch (default) against the two-launch form: one op
    fewer, the same logits up to the stem's summation order, same top-1, golden logits held."""
    m = R.NativeModel("resnet50", state=state50, dtype=dtype)
    try:
        x = np.concatenate([finch, R.weights.generate_input(5, seed=83)])
        m.set_profiling(True)
        fused = m.forward(x, fused=True)
        n_fused = [r["op"] for r in m.profile()]
        m.set_stem_pool_fusion(False)
        plain = m.forward(x, fused=True)
        n_plain = [r["op"] for r in m.profile()]
        m.set_profiling(False)
        assert len(n_plain) == len(n_fused) + 1
        assert "conv2d+epilogue+maxpool" in n_fused and "maxpool2d" not in n_fused
        assert "maxpool2d" in n_plain
        tol = 2e-5 if dtype == "f32" else 0.08
        assert np.abs(fused - plain).max() <= tol
        assert np.array_equal(fused.argmax(1), plain.argmax(1))
        if dtype == "f32":
            want = np.load(os.path.join(golden_dir, "resnet50_finch_logits.npy"))
            assert np.abs(fused[:1] - want).max() <= TOL
        # batch invariance holds for the fused launch too
        m.set_stem_pool_fusion(True)
        assert np.array_equal(m.forward(x[2:3], fused=True), fused[2:3])
        # 2: the launch reads the NCHW input itself -- the layout launch goes, the bits stay
        m.set_stem_pool_fusion(2)
        m.set_profiling(True)
        direct = m.forward(x, fused=True)
        n_direct = [r["op"] for r in m.profile()]
        m.set_profiling(False)
        assert np.array_equal(direct, fused)
        assert len(n_direct) == len(n_fused) - 1 and not any(o.startswith("nchw_to_nhwc") for o in n_direct)
    finally:
        m.close()


def test_chained_conv3_conv1_changes_launches_not_results(state50, finch):
    """bf16 fused mode: conv3 of the 64-channel blocks and conv1 of the block after them as one launch
    (rn_conv_chain_forward_dt, default on) against the separate launches: five ops fewer in
    ResNet-50 (layer1.0's fused conv3 + downsample pair -> 1.1, layer1.1 -> 1.2, layer1.2 -> layer2.0,
    layer2.1 -> 2.2, layer2.2 -> 2.3), nine with the opt-in stage-3 chain (layer3.1 -> 3.2 ... 3.4 -> 3.5),
    the same logits bit for bit, also with two streams and in sub-batches."""
    m = R.NativeModel("resnet50", state=state50, dtype="bf16")
    try:
        x = np.concatenate([finch, R.weights.generate_input(130, seed=61)])
        m.set_streams(1)
        m.set_profiling(True)
        chained = m.forward(x, fused=True)
        ops_chained = [(r["op"], r["layer"]) for r in m.profile()]
        m.set_chain(False)
        plain = m.forward(x, fused=True)
        ops_plain = [(r["op"], r["layer"]) for r in m.profile()]
        m.set_profiling(False)
        assert np.array_equal(chained, plain)
        fused_ops = [l for o, l in ops_chained if o == "conv2d+epilogue+conv2d"]
        assert fused_ops == ["layer1.0.conv3+downsample+next.conv1", "layer1.1.conv3+next.conv1",
                             "layer1.2.conv3+next.conv1", "layer2.1.conv3+next.conv1",
                             "layer2.2.conv3+next.conv1"]
        assert len(ops_plain) == len(ops_chained) + 5
        m.set_chain(True)
        m.set_streams(2)
        assert np.array_equal(m.forward(x, fused=True), plain)
        assert np.array_equal(m.forward(x[:3], fused=True), plain[:3])
    finally:
        m.close()


def test_fp32_chained_conv3_conv1_changes_launches_not_results(state50, finch, golden_dir):
    """fp32 fused mode: the three chains of the 64-channel blocks (layer1.0's pair -> 1.1, layer1.1 -> 1.2,
    layer1.2 -> layer2.0) as one launch each: three ops fewer, the same logits bit for bit, golden
    logits held."""
    m = R.NativeModel("resnet50", state=state50)
    try:
        x = np.concatenate([finch, R.weights.generate_input(70, seed=62)])
        m.set_profiling(True)
        chained = m.forward(x, fused=True)
        ops_chained = [(r["op"], r["layer"]) for r in m.profile()]
        m.set_chain(False)
        plain = m.forward(x, fused=True)
        ops_plain = [(r["op"], r["layer"]) for r in m.profile()]
        m.set_profiling(False)
        assert np.array_equal(chained, plain)
        assert [l for o, l in ops_chained if o == "conv2d+epilogue+conv2d"] == \
            ["layer1.0.conv3+downsample+next.conv1", "layer1.1.conv3+next.conv1", "layer1.2.conv3+next.conv1"]
        assert len(ops_plain) == len(ops_chained) + 3
        want = np.load(os.path.join(golden_dir, "resnet50_finch_logits.npy"))
        assert np.abs(chained[:1] - want).max() <= TOL
    finally:
        m.close()


@pytest.mark.parametrize("arch", ["resnet101", "resnet152"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_chains_in_the_deeper_networks(arch, dtype):
    """The chained launches in ResNet-101 / -152 (the same three of stage 1; bf16: every 128-channel
    boundary of the longer stage 2 as well): the same logits with and without, bit for bit."""
    m = R.NativeModel(arch, state=R.weights.generate_state(arch, seed=3), dtype=dtype)
    try:
        x = R.weights.generate_input(70, seed=4)
        chained = m.forward(x, fused=True)
        m.set_chain(False)
        assert np.array_equal(chained, m.forward(x, fused=True)) and np.isfinite(chained).all()
    finally:
        m.close()


def test_default_stream_split_by_dtype_and_batch(state50):
    """The library's own default of two streams starts where it was measured to pay: parts of 128
    images in fp32 (parts of 96 measured 3 % slower than one stream), parts of 64 with bf16 storage;
    a count the caller sets is taken down to parts of 64 in both.  rn_model_parts says what a forward
    of B images runs as; the bits never depend on it."""
    m = R.NativeModel("resnet50", state=state50)
    try:
        assert m.streams() == 2
        assert [m.parts(B) for B in (1, 64, 128, 192, 255, 256, 512, 2048)] == [1, 1, 1, 1, 1, 2, 2, 2]
        x = R.weights.generate_input(128, seed=5)
        base = m.forward(x, fused=True)          # one stream under the default
        m.set_streams(2)
        assert [m.parts(B) for B in (64, 127, 128, 192, 256)] == [1, 1, 2, 2, 2]
        assert np.array_equal(m.forward(x, fused=True), base)
        m.set_streams(4)
        assert [m.parts(B) for B in (128, 255, 256)] == [2, 2, 4]
        m.set_streams(1)
        assert m.parts(2048) == 1
    finally:
        m.close()
    m = R.NativeModel("resnet50", state=state50, dtype="bf16")
    try:
        assert [m.parts(B) for B in (64, 127, 128, 192, 256)] == [1, 1, 2, 2, 2]
    finally:
        m.close()


def test_two_stream_forward_under_capture_pipeline_and_shards(state50, finch):
    """bf16 models run a batch of >= 128 images as two halves on two streams (fork / join events).
    The same forward captured as a hipGraph (a cross-stream capture), fed through the host
    pipeline, and run by the sharded driver must give the eager forward's bits."""
    m = R.NativeModel("resnet50", state=state50, dtype="bf16")
    try:
        assert m.streams() == 2
        B = 128
        x = R.weights.generate_input(B, seed=97)
        x[100] = finch[0]
        want = m.forward(x, fused=True)
        m.set_streams(1)
        assert np.array_equal(m.forward(x, fused=True), want)
        m.set_streams(2)
        xd = R.FloatTensor.from_numpy(x, R.Device.GPU)
        out = R.FloatTensor((B, 1000), R.Device.GPU)
        g = R.Graph(m, xd.data(), B, out.data(), fused=True)
        R._lib.check(R._lib.lib().rn_memset(m.ctx.handle, out.data(), 0, B * 4000), "memset", m.ctx.handle)
        g.launch(); g.launch(); m.ctx.sync()
        assert np.array_equal(out.numpy(), want)
        g.close()
        pipe = R.Pipeline(m, B, fused=True)
        got = list(pipe.run([x, x[::-1].copy()]))
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want[::-1])
        pipe.close()
    finally:
        m.close()
    sh = R.ShardedModel([0, 0], "resnet50", state=state50, dtype="bf16")
    try:
        logits, top1 = sh.forward(x[:10], fused=True)
        assert np.array_equal(logits, want[:10]) and np.array_equal(top1, want[:10].argmax(1).astype(np.uint64))
    finally:
        sh.close()


def test_sharded_model_equals_whole_batch(state50, model50, finch):
    """rn_shard_*: one host thread + context + model per listed device, contiguous batch split,
    logits concatenated on the host (main.cu:228-254 over several devices).  A
    one-GPU box lists device 0 two and three times: shards must equal the rows of the whole
    batch bit for bit, in image order, and the class indices those of the host argmax."""
    B = 7
    x = R.weights.generate_input(B, seed=55)
    x[4] = finch[0]
    whole = model50.forward(x, fused=True)
    for devices in ([0], [0, 0], [0, 0, 0]):
        g = R.ShardedModel(devices, "resnet50", state=state50)
        try:
            logits, top1 = g.forward(x, fused=True)
            assert np.array_equal(logits, whole), devices
            assert np.array_equal(top1, R.ops.argmax(whole)), devices
            assert int(top1[4]) == 112
            # a batch smaller than the group: the empty shards stay idle
            l1, t1 = g.forward(x[4:5], fused=True)
            assert np.array_equal(l1, whole[4:5]) and int(t1[0]) == 112
            # shard r of the split == rows [lo, hi) of the whole batch, through bench.shard_bounds
            import bench
            for r in range(len(devices)):
                lo, hi = bench.shard_bounds(B, r, len(devices))
                assert (lo, hi) == R.ShardedModel.bounds(B, r, len(devices))
                if hi > lo:
                    assert np.array_equal(model50.forward(x[lo:hi], fused=True), whole[lo:hi])
        finally:
            g.close()
    with pytest.raises(R.RnError):
        R.ShardedModel([99], "resnet50", state=state50)   # no such device: a status, not a crash


def test_plain_c_driver_shards_over_listed_devices(state50, finch, tmp_path):
    """rn_infer --devices 0,0,0 prints the same 'max index is N' lines, in image order, as the
    single-device run of the same batch."""
    wdir = tmp_path / "weights_bin"
    os.mkdir(wdir)
    R.weights.save_weights_bin(state50, str(wdir))
    x = R.weights.generate_input(5, seed=58)
    x[3] = finch[0]
    inp = tmp_path / "batch.bin"
    x.tofile(inp)
    exe = os.path.join(os.path.dirname(R._lib.LIB_PATH), "rn_infer")
    base = [exe, "--arch", "50", "--weights", str(wdir), "--input", str(inp), "--batch", "5"]
    one = subprocess.run(base + ["--device", "0"], capture_output=True, text=True, timeout=300)
    many = subprocess.run(base + ["--devices", "0,0,0"], capture_output=True, text=True, timeout=300)
    assert one.returncode == 0 and many.returncode == 0, one.stderr + many.stderr
    pick = lambda out: [l for l in out.splitlines() if l.startswith("max index is")]
    assert len(pick(one.stdout)) == 5 and pick(one.stdout) == pick(many.stdout)
    assert pick(many.stdout)[3] == "max index is 112"


def test_host_pipeline_matches_plain_forward_bit_exact(model50, finch):
    """rn_pipeline_* (pinned staging, copy stream, two slots in flight) returns, batch by
    batch and in order, exactly what upload -> rn_model_forward -> download returns."""
    B = 3
    batches = [R.weights.generate_input(B, seed=40 + i) for i in range(5)]
    batches[2][1] = finch[0]
    want = [model50.forward(x, fused=True) for x in batches]
    pipe = R.Pipeline(model50, B, fused=True)
    got = list(pipe.run(batches))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    # zero-copy route: fill the staging buffer in place
    buf = pipe.input_buffer()
    buf[...] = batches[4]
    pipe.submit()
    assert pipe.in_flight() == 1
    assert np.array_equal(pipe.collect(), want[4])
    # protocol errors are statuses, not crashes
    with pytest.raises(R.RnError):
        pipe.collect()
    pipe.submit(batches[0]); pipe.submit(batches[1])
    with pytest.raises(R.RnError):
        pipe.submit(batches[2])
    assert np.array_equal(pipe.collect(), want[0])
    assert np.array_equal(pipe.collect(), want[1])
    pipe.close()


def test_resnet101_vs_reference_golden_and_oracle(finch, golden_dir):
    """The third depth the model table knows ([3,4,23,3]; the reference's layer lists are main.cu:109-125 with
    other counts): against the golden logits of the reference's nn.Module classes rebuilt with those counts
    (tests/golden/make_golden.py, round 4: both forward modes, the finch image and the random pair) and against
    the oracle."""
    state = R.weights.generate_state("resnet101", seed=0)
    m = R.NativeModel("resnet101", state=state)
    want = np.load(os.path.join(golden_dir, "resnet101_finch_logits.npy"))
    want2 = np.load(os.path.join(golden_dir, "resnet101_rand2_logits.npy"))
    info = json.load(open(os.path.join(golden_dir, "resnet101_taps.json")))
    rand2 = R.weights.generate_input(2, seed=7)
    try:
        cpu = O.resnet_forward(state, finch, "resnet101")
        for fused in (False, True):
            got = m.forward(finch, fused=fused)
            assert np.abs(got - want).max() <= TOL and np.abs(got - cpu).max() <= TOL
            assert R.ops.argmax(got)[0] == info["finch_top1"] == int(cpu.argmax(1)[0])
            got2 = m.forward(rand2, fused=fused)
            assert np.abs(got2 - want2).max() <= TOL
            assert [int(v) for v in got2.argmax(1)] == info["rand2_top1"]
    finally:
        m.close()


@pytest.mark.parametrize("fused", [False, True])
def test_captured_forward_replays_bit_exact(model50, finch, fused):
    """rn_model_capture: the forward as one hipGraph launch; a replay reads the buffers as they
    are at launch time and gives the eager forward's bits."""
    B = 2
    x0 = R.weights.generate_input(B, seed=61)
    x1 = R.weights.generate_input(B, seed=62)
    x1[1] = finch[0]
    want0, want1 = model50.forward(x0, fused=fused), model50.forward(x1, fused=fused)
    xd = R.FloatTensor.from_numpy(x0, R.Device.GPU)
    out = R.FloatTensor((B, 1000), R.Device.GPU)
    g = R.Graph(model50, xd.data(), B, out.data(), fused=fused)
    assert g.node_count() >= 50
    lib, ctx = R._lib.lib(), model50.ctx
    R._lib.check(lib.rn_memset(ctx.handle, out.data(), 0, B * 4000), "memset", ctx.handle)
    g.launch(); ctx.sync()
    assert np.array_equal(out.numpy(), want0)
    R._lib.check(lib.rn_memcpy_h2d(ctx.handle, xd.data(), x1.ctypes.data, x1.nbytes), "h2d", ctx.handle)
    g.launch(); g.launch(); ctx.sync()
    assert np.array_equal(out.numpy(), want1)
    g.close()
    # capture refuses the modes that put host synchronisation or events inside the forward
    model50.set_profiling(True)
    with pytest.raises(R.RnError):
        R.Graph(model50, xd.data(), B, out.data(), fused=fused)
    model50.set_profiling(False)


def test_pair_fusion_changes_launches_not_results(model50, finch):
    """conv3 + downsample as one contraction (the default) against the two-launch form: four
    launches fewer, same logits within the fold's rounding, same top-1."""
    x = R.weights.generate_input(3, seed=77)
    x[0] = finch[0]
    model50.set_chain(False)   # (the stage-1 pair would otherwise carry the next conv1 as well)
    try:
        with_pair = model50.forward(x, fused=True)
        model50.set_profiling(True)
        model50.forward(x, fused=True)
        n_pair = len(model50.profile())
        model50.set_pair_fusion(False)
        without = model50.forward(x, fused=True)
        n_plain = len(model50.profile())
    finally:
        model50.set_pair_fusion(True)
        model50.set_chain(True)
        model50.set_profiling(False)
    assert n_plain - n_pair == 4
    assert np.abs(with_pair - without).max() <= 2e-5
    assert np.array_equal(with_pair.argmax(1), without.argmax(1))


def test_stem_forms_agree(model50, finch):
    """The fp32 stem as K = 160 (exact form, default) and as K = 224 (4-channel slots) sum the
    same 147 products in different orders."""
    x = R.weights.generate_input(2, seed=88)
    x[1] = finch[0]
    for fused in (True, False):
        a = model50.forward(x, fused=fused)
        model50.set_stem_exact(False)
        try:
            b = model50.forward(x, fused=fused)
        finally:
            model50.set_stem_exact(True)
        assert np.abs(a - b).max() <= 1e-5
        assert np.array_equal(a.argmax(1), b.argmax(1))


def test_split_k_model_latency_mode(model50, finch):
    """Whole network at B = 1 with the K loops of under-filled layers split: logits within the
    reassociation tolerance of the default path, same top-1, deterministic."""
    ctx = model50.ctx
    base = model50.forward(finch, fused=True)
    ctx.set_split_k(16)
    try:
        a = model50.forward(finch, fused=True)
        b = model50.forward(finch, fused=True)
        ops_mode = model50.forward(finch, fused=False)
    finally:
        ctx.set_split_k(0)
    assert np.array_equal(a, b)
    assert np.abs(a - base).max() <= 1e-5 and np.abs(ops_mode - base).max() <= 2e-5
    assert a.argmax(1)[0] == base.argmax(1)[0] == 112
    assert np.array_equal(model50.forward(finch, fused=True), base)


def test_live_graph_pins_arenas_and_scratch(state50):
    """A captured forward holds pointers into the activation arenas and the context scratch: a
    call that would have to reallocate them (a larger batch) is refused until the graph is gone."""
    ctx = R.Context(0)
    m = R.NativeModel("resnet50", state=state50, ctx=ctx)
    try:
        x1 = R.weights.generate_input(1, seed=91)
        x2 = R.weights.generate_input(2, seed=92)
        want1 = m.forward(x1, fused=True)
        xd = R.FloatTensor.from_numpy(x1, R.Device.GPU)   # note: allocated through the default ctx
        out = R.FloatTensor((1, 1000), R.Device.GPU)
        g = R.Graph(m, xd.data(), 1, out.data(), True)
        with pytest.raises(R.RnError):
            m.forward(x2, fused=True)
        g.launch(); ctx.sync()
        assert np.array_equal(out.numpy(), want1)
        g.close()
        assert m.forward(x2, fused=True).shape == (2, 1000)
    finally:
        m.close()
        ctx.close()


def test_two_host_threads_two_contexts(state50):
    """One context (device, stream, scratch) per host thread, no shared mutable state in the
    library (SURVEY.md 8(b) threading): two threads run their halves of a batch at the same
    time and get the bits a single thread gets."""
    import threading
    x = R.weights.generate_input(6, seed=123)
    ref_model = R.NativeModel("resnet50", state=state50)
    want = ref_model.forward(x, fused=True)
    ref_model.close()
    results, errors = [None, None], []

    def work(i):
        try:
            ctx = R.Context(0)
            m = R.NativeModel("resnet50", state=state50, ctx=ctx)
            for _ in range(3):
                results[i] = m.forward(x[3 * i:3 * i + 3], fused=True)
            m.close()
            ctx.close()
        except Exception as e:  # pragma: no cover - surfaced below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert np.array_equal(np.concatenate(results), want)


def test_config3_bf16_b2048_as_eight_shards_on_one_device(state50, model50):
    """BASELINE.json configs[3] end to end, rehearsed on ONE device: ResNet-50 bf16, global batch
    2048 = 8 shards x 256 (rn_shard_* over devices 0 x 8: eight host threads, contexts and models).
    Every shard's rows must be the bits of a single bf16 B=256 forward of the same images; and
    SURVEY 8(d)'s parity bar for this config is measured over ALL 2048 images: top-1 agreement
    with the fp32 engine >= 99 %, with the observed rate, the logit error and the margins of the
    disagreeing images printed (DESIGN.md section 3 quotes them)."""
    B, G = 2048, 8
    x = R.weights.generate_input(B, seed=2048)
    sh = R.ShardedModel([0] * G, "resnet50", state=state50, dtype="bf16")
    try:
        logits, top1 = sh.forward(x, fused=True)
    finally:
        sh.close()
    assert logits.shape == (B, 1000) and np.isfinite(logits).all()
    assert np.array_equal(top1, logits.argmax(1).astype(np.uint64))   # first maximum wins == numpy
    one = R.NativeModel("resnet50", state=state50, dtype="bf16")
    try:
        for r in range(G):
            lo, hi = R.ShardedModel.bounds(B, r, G)
            assert (lo, hi) == (256 * r, 256 * r + 256)
            assert np.array_equal(one.forward(x[lo:hi], fused=True), logits[lo:hi]), r
    finally:
        one.close()
    f32 = np.concatenate([model50.forward(x[i:i + 256], fused=True) for i in range(0, B, 256)])
    agree = logits.argmax(1) == f32.argmax(1)
    rate = float(agree.mean())
    err = np.abs(logits - f32)
    top2 = np.sort(f32, axis=1)[:, -2:]
    margin = top2[:, 1] - top2[:, 0]
    rel = float(np.linalg.norm(logits - f32) / np.linalg.norm(f32))
    print(f"\nconfigs[3] rehearsal: bf16 vs fp32 top-1 agreement {int(agree.sum())}/{B} = {rate:.4%}; "
          f"max|dlogit| {err.max():.4f}, mean {err.mean():.5f}, relative L2 {rel:.3e}; logits range "
          f"[{f32.min():.2f}, {f32.max():.2f}]; fp32 top-2 margins of the disagreeing images: "
          f"{np.sort(margin[~agree])[:12].round(4).tolist()} (median margin of all images {np.median(margin):.3f})")
    assert err.max() <= 0.25
    # a disagreement can only sit where the fp32 margin is below twice the logit error
    assert (margin[~agree] <= 2 * err.max()).all()
    assert rate >= 0.99, rate


def test_sharded_stream_keeps_two_batches_in_flight(state50, model50, finch):
    """rn_shard_stream_*: pinned staging + copy stream + two slots on every device (the composition
    of rn_shard_* with rn_pipeline_*): consecutive batches, two in flight, rows in image order,
    the bits of the plain forward; the staging buffers can be filled in place; protocol errors
    are statuses."""
    B = 10
    xs = [R.weights.generate_input(B, seed=300 + i) for i in range(4)]
    xs[1][7] = finch[0]
    want = [model50.forward(x, fused=True) for x in xs]
    g = R.ShardedModel([0, 0, 0], "resnet50", state=state50)
    try:
        with pytest.raises(R.RnError):
            g.submit(xs[0])                      # not opened yet
        g.stream_open(B, fused=True)
        g.submit(xs[0]); g.submit(xs[1])
        assert g.in_flight() == 2
        with pytest.raises(R.RnError):
            g.submit(xs[2])                      # both slots busy
        with pytest.raises(R.RnError) as e:
            g.forward(xs[3], fused=True)         # the one-shot call is refused while batches are in flight
        assert "in flight" in str(e.value)
        l0, t0 = g.collect()
        assert np.array_equal(l0, want[0]) and np.array_equal(t0, R.ops.argmax(want[0]))
        # zero-copy: write shard r's images into its pinned staging buffer, submit nothing
        for r in range(3):
            buf, lo, hi = g.stream_buffer(r)
            assert (lo, hi) == R.ShardedModel.bounds(B, r, 3) and buf.shape[0] == hi - lo
            buf[...] = xs[2][lo:hi]
        g.submit(None)
        l1, t1 = g.collect()
        assert np.array_equal(l1, want[1]) and int(t1[7]) == 112
        l2, _ = g.collect()
        assert np.array_equal(l2, want[2])
        with pytest.raises(R.RnError):
            g.collect()                          # nothing in flight
        # the one-shot call still works afterwards (it re-sizes the pipelines), and a shard larger
        # than the 128-image chunk goes through in several chunks
        l3, t3 = g.forward(xs[3], fused=True)
        assert np.array_equal(l3, want[3]) and np.array_equal(t3, R.ops.argmax(want[3]))
        g.stream_close()
    finally:
        g.close()
    big = np.concatenate([xs[0]] * 30)[:290]     # one device, 290 images: chunks of 128 + 128 + 34
    g = R.ShardedModel([0], "resnet50", state=state50)
    try:
        lb, tb = g.forward(big, fused=True)
        assert np.array_equal(lb[:10], want[0]) and np.array_equal(lb[280:290], want[0])
        assert np.array_equal(tb, lb.argmax(1).astype(np.uint64))
    finally:
        g.close()


def test_shards_share_one_tuning_pass_and_say_where_they_run(state50, model50):
    """rn_shard_tune: shard 0 measures, the shards with the same share take its table over
    (rn_model_export_tuning / _import_tuning) -- every shard of a node then runs the same tiles, and two
    shards on one device do not sit in each other's timings; a shard with another share (uneven split)
    measures for itself.  With a stream open the tiles are those of a whole shard per launch, not of
    rn_shard_forward's 128-image chunks; refused while submitted batches are in flight.  Tiles never
    change bits.  rn_shard_placement reports each shard's device, NUMA node and the cores its thread got."""
    B = 12
    x = R.weights.generate_input(B, seed=404)
    want = model50.forward(x, fused=True)
    g = R.ShardedModel([0, 0, 0], "resnet50", state=state50)
    try:
        for r in range(3):
            dev, node, cpus = g.placement(r)
            assert dev == 0 and node >= -1
            if cpus:                                 # bound: a list of ranges of CPUs this process may use
                allowed = os.sched_getaffinity(0)
                got = set()
                for part in cpus.split(","):
                    a, _, b = part.partition("-")
                    got |= set(range(int(a), int(b or a) + 1))
                assert got and got <= allowed
        g.tune(x, fused=True)                        # 4 + 4 + 4 images
        t = [g.tuning_of(r) for r in range(3)]
        assert np.array_equal(t[0], t[1]) and np.array_equal(t[0], t[2])
        assert int(t[0][4]) == 4                     # measured at the shard's share
        l, _ = g.forward(x, fused=True)
        assert np.array_equal(l, want)
        g.tune(x[:11], fused=True)                   # 4 + 4 + 3: the last shard measures for itself
        t = [g.tuning_of(r) for r in range(3)]
        assert np.array_equal(t[0], t[1]) and int(t[2][4]) == 3
        g.stream_open(B, fused=True)
        with pytest.raises(R.RnError):
            g.tune(x[:6], fused=True)                # not the open stream's batch
        g.tune(x, fused=True)
        g.submit(x)
        with pytest.raises(R.RnError) as e:
            g.tune(x, fused=True)
        assert "in flight" in str(e.value)
        l, _ = g.collect()
        assert np.array_equal(l, want)
        g.stream_close()
    finally:
        g.close()
    # a table is only taken by the model it was measured for
    a = R.NativeModel("resnet50", state=state50)
    b = R.NativeModel("resnet50", state=state50)
    c = R.NativeModel("resnet50", state=state50, dtype="bf16")
    try:
        with pytest.raises(R.RnError):
            a.export_tuning()                        # not tuned yet
        xd = R.FloatTensor.from_numpy(x, R.Device.GPU)
        out = R.FloatTensor((B, 1000), R.Device.GPU)
        a.tune(xd.data(), B, out.data(), True)
        words = a.export_tuning()
        b.import_tuning(words)
        assert np.array_equal(b.export_tuning(), words)
        assert np.array_equal(b.forward(x, fused=True), want)
        with pytest.raises(R.RnError):
            c.import_tuning(words)                   # another element type
        b.set_pair_fusion(False)
        with pytest.raises(R.RnError):
            b.import_tuning(words)                   # another set of launches
        bad = words.copy(); bad[12] = 999            # a candidate this build does not have
        b.set_pair_fusion(True)
        with pytest.raises(R.RnError):
            b.import_tuning(bad)
    finally:
        for m in (a, b, c):
            m.close()


def test_pipeline_ragged_batch_and_class_indices(model50, finch):
    """rn_pipeline_submit_n / _collect_n: a last batch with fewer images than the pipeline's B, and
    the class indices (first maximum wins, main.cu:243-249) next to the logits."""
    B = 4
    x = R.weights.generate_input(B, seed=77)
    x[1] = finch[0]
    want = model50.forward(x, fused=True)
    pipe = R.Pipeline(model50, B, fused=True)
    try:
        pipe.submit(x); pipe.submit(x[:3])
        l0, t0 = pipe.collect_top1()
        l1, t1 = pipe.collect_top1()
        assert np.array_equal(l0, want) and np.array_equal(t0, R.ops.argmax(want)) and int(t0[1]) == 112
        assert l1.shape == (3, 1000) and np.array_equal(l1, want[:3]) and np.array_equal(t1, t0[:3])
        with pytest.raises(R.RnError):
            pipe.submit(np.concatenate([x, x]))   # more images than the slots hold
    finally:
        pipe.close()


def test_live_graph_pins_the_scratch_of_every_stream(state50):
    """With the batch split over several streams the captured nodes also point into the scratch of
    the model's secondary contexts: a later eager forward that would have to grow one of those
    (fewer, larger parts at the same B) must be refused -- or leave the replay's bits alone --
    never free memory the graph still uses."""
    ctx = R.Context(0)
    m = R.NativeModel("resnet50", state=state50, ctx=ctx)
    try:
        B = 256
        m.set_streams(4)
        x = R.weights.generate_input(B, seed=17)
        want = m.forward(x, fused=True)
        xd = R.FloatTensor.from_numpy(x, R.Device.GPU)
        out = R.FloatTensor((B, 1000), R.Device.GPU)
        g = R.Graph(m, xd.data(), B, out.data(), True)
        m.set_streams(2)                 # parts of 128 instead of 64 on the second context
        try:
            eager = m.forward(x, fused=True)
            assert np.array_equal(eager, want)
        except R.RnError:
            pass                         # refused: the pinned scratch would have had to grow
        R._lib.check(R._lib.lib().rn_memset(ctx.handle, out.data(), 0, B * 4000), "memset", ctx.handle)
        g.launch(); ctx.sync()
        assert np.array_equal(out.numpy(), want)
        g.close()
        assert np.array_equal(m.forward(x, fused=True), want)   # unpinned again
    finally:
        m.close()
        ctx.close()


def test_model_outlives_its_graphs(state50):
    """C teardown order (rn_hip.h): a graph captured from a model that runs its batch as parts on several
    streams pins the contexts the MODEL owns for those streams; rn_model_destroy frees them.  Destroying
    the model first must be refused (RN_ERR_INVALID, nothing freed) -- rn_graph_destroy would otherwise
    write into freed contexts -- and must work once the graph is gone."""
    lib = R._lib.lib()
    ctx = R.Context(0)
    m = R.NativeModel("resnet50", state=state50, ctx=ctx, dtype="bf16")
    B = 128
    m.set_streams(2)
    assert m.parts(B) == 2
    x = R.weights.generate_input(B, seed=19)
    want = m.forward(x, fused=True)
    xd = R.FloatTensor.from_numpy(x, R.Device.GPU)
    out = R.FloatTensor((B, 1000), R.Device.GPU)
    g = R.Graph(m, xd.data(), B, out.data(), True)
    g2 = R.Graph(m, xd.data(), B, out.data(), True)
    assert lib.rn_model_destroy(m.handle) == R._lib.RN_ERR_INVALID      # the raw C call, model first
    with pytest.raises(R.RnError):
        m.close()
    g.launch(); ctx.sync()                                               # everything is still there
    assert np.array_equal(out.numpy(), want)
    g.close()
    assert lib.rn_model_destroy(m.handle) == R._lib.RN_ERR_INVALID      # one graph left
    g2.launch(); ctx.sync()
    g2.close()
    m.close()                                                            # now it goes
    assert m.handle is None
    ctx.close()


def test_plain_c_driver_names_the_fixed_geometry(state50, tmp_path):
    """rn_model_forward has no size argument (224 x 224 like main.cu:230): rn_infer checks the file's
    element count and says what the driver takes instead of reading a wrong-sized image."""
    wdir = tmp_path / "weights_bin"
    os.mkdir(wdir)
    R.weights.save_weights_bin(state50, str(wdir))
    inp = tmp_path / "small.bin"
    R.weights.generate_input(1, seed=1, hw=160).tofile(inp)
    exe = os.path.join(os.path.dirname(R._lib.LIB_PATH), "rn_infer")
    for extra in (["--device", "0"], ["--devices", "0,0"]):
        r = subprocess.run([exe, "--arch", "50", "--weights", str(wdir), "--input", str(inp), "--batch", "1"] + extra,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "3 x 224 x 224" in r.stderr and "unsupported" in r.stderr, r.stderr


def test_sharded_stream_with_idle_shards_and_teardown_in_flight(state50, model50, finch):
    """A batch smaller than the group leaves shards without images: they stay idle through open /
    submit / collect and have no staging buffer.  Destroying a group (and a pipeline) with batches
    still in flight must wait for them, not crash."""
    x = R.weights.generate_input(2, seed=808)
    x[1] = finch[0]
    want = model50.forward(x, fused=True)
    g = R.ShardedModel([0, 0, 0], "resnet50", state=state50)
    try:
        g.stream_open(2, fused=True)
        bufs = [g.stream_buffer(r) for r in range(3)]
        assert [b[1:] for b in bufs] == [(0, 1), (1, 2), (2, 2)] and bufs[2][0] is None
        g.submit(x)
        logits, top1 = g.collect()
        assert np.array_equal(logits, want) and int(top1[1]) == 112
        g.submit(x); g.submit(x)          # left in flight on purpose
    finally:
        g.close()
    pipe = R.Pipeline(model50, 2, fused=True)
    pipe.submit(x); pipe.submit(x[:1])
    pipe.close()                          # two batches in flight
    assert np.array_equal(model50.forward(x, fused=True), want)
