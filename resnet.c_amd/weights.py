"""Weight files either side of the forward path.

Two things live here:

* the on-disk format the reference's tools write and its loader reads:
  one headerless native-endian fp32 file per ``state_dict`` key inside a
  ``weights_bin/`` directory (reference ``save_weights.py:8-12`` writes them,
  ``cuda/tensor.cuh:126-147`` reads ``file_size / 4`` floats back,
  ``cuda/nn.cuh:21,58-61,113,117`` builds the file names);
* a deterministic synthetic generator.  Pretrained weights cannot be fetched
  offline, so parity and throughput runs use weights produced from a
  counter-based hash: the value of element ``i`` of tensor ``name`` depends on
  ``(seed, name, i)`` only, is computed in float64 and rounded once to fp32,
  so every host produces bit-identical tensors.

The layer table follows the reference's model factory
(``cuda/inference/main.cu:53-89,109-125``): bottleneck blocks with the stride
on the 3x3 convolution and a projection shortcut on block 0 of every stage.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, List, Tuple

import numpy as np

# stage widths (in, mid, out) and strides are the same for every depth
# (main.cu:116-119); only the block counts differ.
STAGE_WIDTHS = ((64, 64, 256), (256, 128, 512), (512, 256, 1024), (1024, 512, 2048))
STAGE_STRIDES = (1, 2, 2, 2)
DEPTHS = {
    "resnet50": (3, 4, 6, 3),
    "resnet101": (3, 4, 23, 3),
    "resnet152": (3, 8, 36, 3),  # main.cu:116-119
}
NUM_CLASSES = 1000
BN_FIELDS = ("weight", "bias", "running_mean", "running_var")


def depths_of(arch: str) -> Tuple[int, int, int, int]:
    try:
        return DEPTHS[arch]
    except KeyError:
        raise ValueError(f"unknown arch {arch!r}; expected one of {sorted(DEPTHS)}") from None


def conv_specs(arch: str) -> List[Tuple[str, int, int, int, int, int]]:
    """(name, cin, cout, k, stride, pad) for every convolution, in forward order."""
    out = [("conv1", 3, 64, 7, 2, 3)]
    for li, ((cin, mid, cout), stride, n) in enumerate(
        zip(STAGE_WIDTHS, STAGE_STRIDES, depths_of(arch)), start=1
    ):
        for bi in range(n):
            pre = f"layer{li}.{bi}"
            b_in = cin if bi == 0 else cout
            b_stride = stride if bi == 0 else 1
            if bi == 0 and (b_stride != 1 or b_in != cout):
                out.append((f"{pre}.downsample.0", b_in, cout, 1, b_stride, 0))
            out.append((f"{pre}.conv1", b_in, mid, 1, 1, 0))
            out.append((f"{pre}.conv2", mid, mid, 3, b_stride, 1))
            out.append((f"{pre}.conv3", mid, cout, 1, 1, 0))
    return out


def bn_of(conv_name: str) -> str:
    """Name of the batch-norm that follows a convolution (main.cu:59-75,111-112)."""
    if conv_name.endswith("downsample.0"):
        return conv_name[:-1] + "1"
    head, _, tail = conv_name.rpartition("conv")
    return f"{head}bn{tail}"


def tensor_specs(arch: str) -> List[Tuple[str, Tuple[int, ...]]]:
    """Every file the loader reads: (state_dict key, shape)."""
    specs: List[Tuple[str, Tuple[int, ...]]] = []
    for name, cin, cout, k, _s, _p in conv_specs(arch):
        specs.append((f"{name}.weight", (cout, cin, k, k)))
        bn = bn_of(name)
        for f in BN_FIELDS:
            specs.append((f"{bn}.{f}", (cout,)))
    specs.append(("fc.weight", (NUM_CLASSES, 2048)))
    specs.append(("fc.bias", (NUM_CLASSES,)))
    return specs


def param_count(arch: str) -> int:
    """Learnable parameters (running stats excluded), e.g. 25,557,032 for resnet50."""
    n = 0
    for key, shape in tensor_specs(arch):
        if key.endswith("running_mean") or key.endswith("running_var"):
            continue
        n += int(np.prod(shape))
    return n


# --------------------------------------------------------------------------
# counter-based generator
# --------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def uniform01(name: str, n: int, seed: int, offset: int = 0) -> np.ndarray:
    """n float64 values in [0,1) with 24 random bits each (exact in fp32)."""
    key = (_fnv1a64(name) ^ ((seed * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        ctr = idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(key)
    bits = _splitmix64(ctr) >> np.uint64(40)
    return bits.astype(np.float64) * (1.0 / 16777216.0)


def _uniform(name: str, shape, lo: float, hi: float, seed: int) -> np.ndarray:
    n = int(np.prod(shape))
    u = uniform01(name, n, seed)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def generate_tensor(key: str, shape, seed: int = 0) -> np.ndarray:
    """Synthetic tensor for one state_dict key.

    conv: He-uniform over fan_in; BN gamma in [0.5,1.5), beta and running mean in
    [-0.1,0.1), running var in [0.5,1.5); fc uniform(+-1/sqrt(in)).  The last
    batch-norm of every block (bn3) gets a smaller gamma so the residual stream
    of the 50-block network stays O(1).
    """
    if key.endswith("running_var"):
        return _uniform(key, shape, 0.5, 1.5, seed)
    if key.endswith("running_mean"):
        return _uniform(key, shape, -0.1, 0.1, seed)
    if key.startswith("fc."):
        bound = 1.0 / np.sqrt(2048.0)
        return _uniform(key, shape, -bound, bound, seed)
    if key.endswith(".bias"):
        return _uniform(key, shape, -0.1, 0.1, seed)
    if len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        bound = float(np.sqrt(6.0 / fan_in))
        return _uniform(key, shape, -bound, bound, seed)
    if ".bn3." in key:
        return _uniform(key, shape, 0.02, 0.1, seed)
    return _uniform(key, shape, 0.5, 1.5, seed)


def generate_state(arch: str, seed: int = 0) -> Dict[str, np.ndarray]:
    return {k: generate_tensor(k, s, seed) for k, s in tensor_specs(arch)}


def generate_input(batch: int, seed: int = 0, hw: int = 224, name: str = "input") -> np.ndarray:
    """[batch,3,hw,hw] NCHW fp32, uniform in [-2,2): image i depends on (seed, i) only."""
    per = 3 * hw * hw
    out = np.empty((batch, 3, hw, hw), dtype=np.float32)
    for i in range(batch):
        u = uniform01(name, per, seed, offset=i * per)
        out[i] = (-2.0 + 4.0 * u).astype(np.float32).reshape(3, hw, hw)
    return out


# --------------------------------------------------------------------------
# weights_bin/ directory format
# --------------------------------------------------------------------------
def save_weights_bin(state: Dict[str, np.ndarray], dir_name: str) -> None:
    """Write one raw fp32 file per key, like reference save_weights.py:8-12."""
    os.makedirs(dir_name, exist_ok=True)
    for key, arr in state.items():
        np.ascontiguousarray(arr, dtype=np.float32).tofile(os.path.join(dir_name, key))


def load_weights_bin(arch: str, dir_name: str) -> Dict[str, np.ndarray]:
    """Read the files the reference loader reads; other files (e.g.
    ``*.num_batches_tracked``, which an export contains) are ignored."""
    state = {}
    for key, shape in tensor_specs(arch):
        path = os.path.join(dir_name, key)
        arr = np.fromfile(path, dtype=np.float32)
        want = int(np.prod(shape))
        if arr.size != want:
            raise ValueError(f"{path}: {arr.size} floats on disk, expected {want} for {shape}")
        state[key] = arr.reshape(shape)
    return state


def iter_blocks(arch: str) -> Iterator[Tuple[str, int, int, int, int, bool]]:
    """(prefix, cin, mid, cout, stride, has_downsample) per bottleneck block."""
    for li, ((cin, mid, cout), stride, n) in enumerate(
        zip(STAGE_WIDTHS, STAGE_STRIDES, depths_of(arch)), start=1
    ):
        for bi in range(n):
            b_in = cin if bi == 0 else cout
            b_stride = stride if bi == 0 else 1
            yield (f"layer{li}.{bi}", b_in, mid, cout, b_stride,
                   bi == 0 and (b_stride != 1 or b_in != cout))
