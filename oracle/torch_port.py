"""torch.nn.functional port of the reference's PyTorch model (CPU).

TEST INFRASTRUCTURE ONLY -- the CPU baseline timed by bench.py and a second
checker for tests; never imported by the product package.

Restates the graph of reference pytorch_inference.py:
    ResnetBlock.forward   :66-82   (conv-bn-relu x2, conv-bn, += shortcut, relu)
    make_layer            :85-110  (projection shortcut on block 0)
    Resnet152.forward     :135-162 (stem, maxpool, 4 stages, adaptive avgpool, fc)
with the block counts as a parameter.  Inference-mode batch-norm
(``training=False``, eps 1e-5 = nn.BatchNorm2d default).  It is pinned to the
reference by tests/golden/*_logits.npy, which were produced by the reference's
own nn.Module classes on the same generated weights (tests/golden/make_golden.py).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

_DEPTHS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3)}


def to_torch(state: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in state.items()}


def _bn(t, name, x):
    return F.batch_norm(x, t[name + ".running_mean"], t[name + ".running_var"],
                        t[name + ".weight"], t[name + ".bias"], training=False, eps=1e-5)


def _block(t, pre, x, stride, has_ds):
    shortcut = x
    if has_ds:
        shortcut = _bn(t, pre + ".downsample.1",
                       F.conv2d(x, t[pre + ".downsample.0.weight"], stride=stride))
    y = F.relu(_bn(t, pre + ".bn1", F.conv2d(x, t[pre + ".conv1.weight"])))
    y = F.relu(_bn(t, pre + ".bn2",
                   F.conv2d(y, t[pre + ".conv2.weight"], stride=stride, padding=1)))
    y = _bn(t, pre + ".bn3", F.conv2d(y, t[pre + ".conv3.weight"]))
    y = y + shortcut
    return F.relu(y)


@torch.no_grad()
def resnet_forward(t: Dict[str, torch.Tensor], x: torch.Tensor, arch: str = "resnet50"):
    y = F.conv2d(x, t["conv1.weight"], stride=2, padding=3)
    y = F.relu(_bn(t, "bn1", y))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    for li, (n, stride) in enumerate(zip(_DEPTHS[arch], (1, 2, 2, 2)), start=1):
        for bi in range(n):
            y = _block(t, f"layer{li}.{bi}", y, stride if bi == 0 else 1, bi == 0)
    y = F.adaptive_avg_pool2d(y, (1, 1))
    return F.linear(y.flatten(-3), t["fc.weight"], t["fc.bias"])
