"""The CPU checker against the reference's own outputs (no GPU)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import resnet_c_amd as R
from oracle import oracle as O
from oracle import torch_port as TP


def test_kat_reference_test_patterns(golden_dir):
    """Input patterns of the reference's cuda/test.cu; expected values from
    torch.nn.functional, conv rows as recorded in SURVEY.md section 4."""
    k = np.load(os.path.join(golden_dir, "ops_kat.npz"))
    y = O.conv2d(k["conv_x"], k["conv_w"])
    assert y.shape == (2, 2, 6, 6)
    assert np.array_equal(y, k["conv_y"])
    assert y[0, 0, 0].tolist() == [39, 45, 51, 57, 63, 69]
    assert y[0, 1, 0].tolist() == [103, 125, 147, 169, 191, 213]
    assert np.array_equal(O.linear(k["lin_x"], k["lin_w"], k["lin_b"]), k["lin_y"])
    assert np.array_equal(O.relu(k["relu_x"]), k["relu_y"])


@pytest.mark.parametrize("arch", ["resnet50", "resnet101", "resnet152"])
def test_oracle_reproduces_reference_logits(arch, golden_dir, finch):
    """Golden logits were produced by the reference's nn.Module classes
    (tests/golden/make_golden.py).  Tolerance 1e-4 = the north star's; observed ~5e-6."""
    state = R.weights.generate_state(arch, seed=0)
    got = O.resnet_forward(state, finch, arch)
    want = np.load(os.path.join(golden_dir, f"{arch}_finch_logits.npy"))
    want64 = np.load(os.path.join(golden_dir, f"{arch}_finch_logits_f64.npy"))
    info = json.load(open(os.path.join(golden_dir, f"{arch}_taps.json")))
    assert np.abs(got - want).max() <= 1e-4
    assert np.abs(got - want64).max() <= 1e-4
    assert O.argmax(got)[0] == info["finch_top1"] == int(want.argmax(1)[0])
    assert info["finch_top2_gap_f64"] > 1e-2  # top-1 is not a coin flip


def test_torch_port_reproduces_reference_logits(golden_dir, finch, state50):
    t = TP.to_torch(state50)
    got = TP.resnet_forward(t, torch.from_numpy(finch), "resnet50").numpy()
    want = np.load(os.path.join(golden_dir, "resnet50_finch_logits.npy"))
    assert np.abs(got - want).max() <= 1e-5
    rand2 = R.weights.generate_input(2, seed=7)
    got2 = TP.resnet_forward(t, torch.from_numpy(rand2), "resnet50").numpy()
    assert np.abs(got2 - np.load(os.path.join(golden_dir, "resnet50_rand2_logits.npy"))).max() <= 1e-5


CONV_CASES = [
    # B, Cin, Cout, H, W, k, stride, pad
    (1, 3, 8, 9, 11, 7, 2, 3),
    (2, 5, 4, 8, 8, 3, 1, 1),
    (2, 4, 6, 7, 9, 3, 2, 1),
    (1, 8, 8, 5, 5, 1, 1, 0),
    (3, 8, 4, 6, 6, 1, 2, 0),
    (1, 2, 3, 4, 4, 2, 1, 0),
    (1, 1, 1, 3, 3, 3, 1, 2),  # more padding than data on each side
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_against_torch(case):
    B, Cin, Cout, H, W, k, s, p = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, H, W), dtype=np.float32)
    w = rng.standard_normal((Cout, Cin, k, k), dtype=np.float32)
    want = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), stride=s, padding=p).numpy()
    got = O.conv2d(x, w, s, p)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_pools_bn_against_torch():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 6, 9, 9), dtype=np.float32)
    tx = torch.from_numpy(x)
    assert np.array_equal(O.maxpool2d(x, 3, 2, 1), F.max_pool2d(tx, 3, 2, 1).numpy())
    np.testing.assert_allclose(O.avgpool2d(x, 3, 2, 1),
                               F.avg_pool2d(tx, 3, 2, 1, count_include_pad=True).numpy(),
                               rtol=1e-6, atol=1e-6)
    x7 = rng.standard_normal((2, 6, 7, 7), dtype=np.float32)
    np.testing.assert_allclose(O.avgpool2d(x7, 7)[:, :, 0, 0], x7.mean((2, 3)), rtol=1e-5, atol=1e-6)
    w, b = rng.random(6, dtype=np.float32) + 0.5, rng.standard_normal(6, dtype=np.float32)
    m, v = rng.standard_normal(6, dtype=np.float32), rng.random(6, dtype=np.float32) + 0.5
    want = F.batch_norm(tx, torch.from_numpy(m), torch.from_numpy(v), torch.from_numpy(w),
                        torch.from_numpy(b), training=False, eps=1e-5).numpy()
    np.testing.assert_allclose(O.batchnorm2d(x, w, b, m, v), want, rtol=1e-5, atol=1e-6)


def test_edge_semantics():
    # fmax(NaN, 0) == 0 (reference relu, ops.cu:136); in-place add; first-max-wins argmax
    x = np.array([np.nan, -1.0, 2.0, -0.0], dtype=np.float32)
    assert O.relu(x).tolist() == [0.0, 0.0, 2.0, 0.0]
    a = np.arange(5, dtype=np.float32)
    assert O.add_(a, np.ones(5, dtype=np.float32)).tolist() == [1, 2, 3, 4, 5]
    logits = np.zeros((3, 10), dtype=np.float32)
    logits[0, [3, 7]] = 5.0  # tie -> first
    logits[1, 0] = np.nan    # NaN at index 0 is never displaced ('<' is false)
    logits[2, 4] = np.nan
    logits[2, 6] = 1.0
    assert O.argmax(logits).tolist() == [3, 0, 6]
    assert O.conv_output_size(224, 7, 2, 3) == 112
    assert O.conv_output_size(112, 3, 2, 1) == 56
    assert O.conv_output_size(7, 7, 1, 0) == 1
