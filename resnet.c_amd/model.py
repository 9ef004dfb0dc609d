"""The model graph, twice.

``ResnetModel`` / ``createResnet`` / ``resnetForward`` mirror the reference
driver cuda/inference/main.cu:7-226 object for object on top of the nn layer
wrappers (one C-ABI call per reference op, NCHW tensors, lazily cached
activations) -- the literal drop-in for code written against ops/nn/tensor.

``NativeModel`` wraps the plain-C driver inside librn_hip.so (rn_model.c): NHWC
engine layout, packed weights, optional fused epilogues, no Python in the loop.
It is what bench.py times.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib as L
from . import nn, weights
from .tensor import Context, Device, FloatTensor, Shape, get_ctx

ARCH_ID = {"resnet50": 50, "resnet101": 101, "resnet152": 152}


# ---------------------------------------------------------------------------
# reference-shaped graph (main.cu)
# ---------------------------------------------------------------------------
class Downsample:  # main.cu:7-16
    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d):
        self.conv, self.bn = conv, bn
        self.act = FloatTensor(Device.GPU)


class ResnetBlock:  # main.cu:18-46
    def __init__(self, conv1, bn1, conv2, bn2, conv3, bn3, downsample: Optional[Downsample],
                 in_channels, inter_channels, out_channels, stride):
        self.conv1, self.bn1 = conv1, bn1
        self.conv2, self.bn2 = conv2, bn2
        self.conv3, self.bn3 = conv3, bn3
        self.downsample = downsample
        self.in_channels, self.inter_channels = in_channels, inter_channels
        self.out_channels, self.stride = out_channels, stride
        self.act1_out = FloatTensor(Device.GPU)
        self.act2_out = FloatTensor(Device.GPU)
        self.act3_out = FloatTensor(Device.GPU)


class Layer:  # main.cu:48-51
    def __init__(self, blocks: List[ResnetBlock]):
        self.blocks = blocks


class _Loader:
    """Where layer weights come from: a weights_bin/ directory (the reference's only
    source, nn.cuh:18-25) or an in-memory state dict."""

    def __init__(self, state: Optional[Dict[str, np.ndarray]] = None):
        self.state = state

    def tensor(self, key: str, shape) -> FloatTensor:
        if self.state is None:
            return FloatTensor.loadToCuda(nn.WEIGHTS_DIR + key).view(Shape(shape))
        return FloatTensor.from_numpy(self.state[key], Device.GPU).view(Shape(shape))

    def conv(self, name, cin, cout, k, stride=1, padding=0) -> nn.Conv2d:
        return nn.Conv2d(self.tensor(name + ".weight", (cout, cin, k, k)), cin, cout, k, stride,
                         padding)

    def bn(self, name, c) -> nn.BatchNorm2d:
        return nn.BatchNorm2d(*(self.tensor(f"{name}.{f}", (c,)) for f in weights.BN_FIELDS), c)

    def linear(self, name, fin, fout) -> nn.Linear:
        return nn.Linear(self.tensor(name + ".weight", (fout, fin)),
                         self.tensor(name + ".bias", (fout,)), fin, fout)


def createLayer(ld: _Loader, layer_id: int, in_channels: int, inter_channels: int,
                out_channels: int, n_blocks: int, stride: int = 1) -> Layer:
    """main.cu:53-89."""

    def load_block(block_id, cin, mid, cout, s) -> ResnetBlock:
        pre = f"layer{layer_id}.{block_id}."
        ds = None
        if block_id == 0 and (s != 1 or cin != cout):  # main.cu:71
            ds = Downsample(ld.conv(pre + "downsample.0", cin, cout, 1, s),
                            ld.bn(pre + "downsample.1", cout))
        return ResnetBlock(ld.conv(pre + "conv1", cin, mid, 1), ld.bn(pre + "bn1", mid),
                           ld.conv(pre + "conv2", mid, mid, 3, s, 1), ld.bn(pre + "bn2", mid),
                           ld.conv(pre + "conv3", mid, cout, 1), ld.bn(pre + "bn3", cout),
                           ds, cin, mid, cout, s)

    blocks = [load_block(0, in_channels, inter_channels, out_channels, stride)]
    for i in range(1, n_blocks):
        blocks.append(load_block(i, out_channels, inter_channels, out_channels, 1))
    return Layer(blocks)


class ResnetModel:  # main.cu:91-107
    def __init__(self, ld: _Loader, arch: str):
        d = weights.depths_of(arch)
        self.arch = arch
        self.conv1 = ld.conv("conv1", 3, 64, 7, 2, 3)
        self.bn1 = ld.bn("bn1", 64)
        self.act1_out = FloatTensor(Device.GPU)
        self.maxpool = nn.Pool2d(64, 3, 2, 1)
        self.maxpool_out = FloatTensor(Device.GPU)
        self.layer1 = createLayer(ld, 1, 64, 64, 256, d[0])
        self.layer2 = createLayer(ld, 2, 256, 128, 512, d[1], 2)
        self.layer3 = createLayer(ld, 3, 512, 256, 1024, d[2], 2)
        self.layer4 = createLayer(ld, 4, 1024, 512, 2048, d[3], 2)
        self.avgpool = nn.Pool2d(2048, 7)
        self.avgpool_out = FloatTensor(Device.GPU)
        self.fc = ld.linear("fc", 2048, 1000)
        self.fc_out = FloatTensor(Device.GPU)


def createResnet(arch: str = "resnet152", state: Optional[Dict[str, np.ndarray]] = None):
    """createResnet152 (main.cu:109-125) for any depth; weights from ``weights_bin/``
    (state=None, like the reference) or from a dict."""
    return ResnetModel(_Loader(state), arch)


def createResnet152(state=None) -> ResnetModel:
    return createResnet("resnet152", state)


def _ensure(holder, attr: str, shape: Shape) -> FloatTensor:
    """`if (!act) act = FloatTensor(shape, GPU)` (main.cu:141-143 and friends)."""
    t = getattr(holder, attr)
    if not t or t.shape() != shape:
        t = FloatTensor(shape, Device.GPU)
        setattr(holder, attr, t)
    return t


def layerForward(layer: Layer, x: FloatTensor) -> FloatTensor:
    """main.cu:127-166."""
    y = x
    for block in layer.blocks:
        if block.downsample:
            ds = block.downsample
            _ensure(ds, "act", ds.conv.getOutShape(y.shape()))
            ds.conv.forward(y, ds.act)
            ds.bn.forward(ds.act, ds.act)
        _ensure(block, "act1_out", block.conv1.getOutShape(y.shape()))
        block.conv1.forward(y, block.act1_out)
        block.bn1.forward(block.act1_out, block.act1_out)
        nn.reluForward(block.act1_out, block.act1_out)

        _ensure(block, "act2_out", block.conv2.getOutShape(block.act1_out.shape()))
        block.conv2.forward(block.act1_out, block.act2_out)
        block.bn2.forward(block.act2_out, block.act2_out)
        nn.reluForward(block.act2_out, block.act2_out)

        _ensure(block, "act3_out", block.conv3.getOutShape(block.act2_out.shape()))
        block.conv3.forward(block.act2_out, block.act3_out)
        block.bn3.forward(block.act3_out, block.act3_out)
        nn.addForward(block.act3_out, block.downsample.act if block.downsample else y,
                      block.act3_out)
        nn.reluForward(block.act3_out, block.act3_out)
        y = block.act3_out
    return y


def resnetForward(model: ResnetModel, x: FloatTensor) -> FloatTensor:
    """resnet152Forward (main.cu:168-226) without the progress prints and without the
    unused D2H copy of the layer4 activation (main.cu:207).  Returns model.fc_out."""
    assert x.device == Device.GPU
    x.shape().as_tuple(4)
    conv1_out_shape = model.conv1.getOutShape(x.shape())
    _ensure(model, "act1_out", conv1_out_shape)
    model.conv1.forward(x, model.act1_out)
    model.bn1.forward(model.act1_out, model.act1_out)
    nn.reluForward(model.act1_out, model.act1_out)
    _ensure(model, "maxpool_out", model.maxpool.getOutShape(conv1_out_shape))
    model.maxpool.maxforward(model.act1_out, model.maxpool_out)
    y = layerForward(model.layer1, model.maxpool_out)
    y = layerForward(model.layer2, y)
    y = layerForward(model.layer3, y)
    y = layerForward(model.layer4, y)
    _ensure(model, "avgpool_out", model.avgpool.getOutShape(y.shape()))
    model.avgpool.avgforward(y, model.avgpool_out)
    s = model.avgpool_out.shape()
    flat = model.avgpool_out.view(Shape((s[0], s[1] * s[2] * s[3])))
    _ensure(model, "fc_out", Shape((flat.shape()[0], model.fc.out_features)))
    model.fc.forward(flat, model.fc_out)
    return model.fc_out


def argmax(logits: np.ndarray) -> np.ndarray:
    """Host argmax of main.cu:243-251: strict '<', first maximum wins."""
    out = np.zeros(logits.shape[0], dtype=np.int64)
    for b in range(logits.shape[0]):
        mx = 0
        row = logits[b]
        for i in range(1, row.shape[0]):
            if row[mx] < row[i]:
                mx = i
        out[b] = mx
    return out


# ---------------------------------------------------------------------------
# the C driver
# ---------------------------------------------------------------------------
class NativeModel:
    def __init__(self, arch: str = "resnet50", state: Optional[Dict[str, np.ndarray]] = None,
                 weights_dir: Optional[str] = None, ctx: Optional[Context] = None,
                 dtype: str = "f32"):
        self.ctx = ctx or get_ctx()
        self.arch = arch
        lib = L.lib()
        h = ctypes.c_void_p()
        L.check(lib.rn_model_create(self.ctx.handle, ctypes.byref(h), ARCH_ID[arch]),
                "rn_model_create", self.ctx.handle)
        self.handle = h
        if (state is None) == (weights_dir is None):
            raise ValueError("give exactly one of state / weights_dir")
        if weights_dir is not None:
            L.check(lib.rn_model_load_dir(h, weights_dir.encode()), "rn_model_load_dir",
                    self.ctx.handle)
        else:
            for key, numel in self.tensor_keys():
                arr = np.ascontiguousarray(state[key], dtype=np.float32)
                assert arr.size == numel, (key, arr.size, numel)
                L.check(lib.rn_model_set_tensor(h, key.encode(), arr.ctypes.data, numel),
                        f"rn_model_set_tensor({key})", self.ctx.handle)
        self.dtype = dtype
        L.check(lib.rn_model_set_dtype(h, {"f32": L.RN_DTYPE_F32, "bf16": L.RN_DTYPE_BF16}[dtype]),
                "rn_model_set_dtype", self.ctx.handle)
        L.check(lib.rn_model_finalize(h), "rn_model_finalize", self.ctx.handle)

    def tensor_keys(self) -> List[Tuple[str, int]]:
        out, i, n = [], 0, ctypes.c_uint64()
        while True:
            k = L.lib().rn_model_tensor_key(self.handle, i, ctypes.byref(n))
            if k is None:
                return out
            out.append((k.decode(), n.value))
            i += 1

    def forward_ptr(self, input_ptr: int, B: int, logits_ptr: int, fused: bool = True) -> None:
        """Queue one forward on the context's stream (asynchronous)."""
        L.check(L.lib().rn_model_forward(self.handle, input_ptr, B, logits_ptr,
                                         L.RN_FWD_FUSED if fused else L.RN_FWD_REFERENCE_OPS),
                "rn_model_forward", self.ctx.handle)

    def forward(self, x: np.ndarray, fused: bool = True) -> np.ndarray:
        """NCHW host array -> logits host array (synchronous convenience)."""
        xin = FloatTensor.from_numpy(x, Device.GPU)
        B = x.shape[0]
        out = FloatTensor((B, 1000), Device.GPU)
        self.forward_ptr(xin.data(), B, out.data(), fused)
        self.ctx.sync()
        return out.numpy()

    def tune(self, input_ptr: int, B: int, logits_ptr: int, fused: bool = True) -> None:
        """Pick the fastest contraction tile per layer for batch B (results unchanged)."""
        L.check(L.lib().rn_model_tune(self.handle, input_ptr, B, logits_ptr,
                                      L.RN_FWD_FUSED if fused else L.RN_FWD_REFERENCE_OPS),
                "rn_model_tune", self.ctx.handle)

    def export_tuning(self) -> np.ndarray:
        """The tile table of the last tune() as uint64 words (rn_model_export_tuning): for a model of the
        same architecture, element type and settings on an identical device, or a later process."""
        n = ctypes.c_uint64()
        L.check(L.lib().rn_model_export_tuning(self.handle, None, 0, ctypes.byref(n)), "rn_model_export_tuning")
        words = np.zeros(n.value, dtype=np.uint64)
        L.check(L.lib().rn_model_export_tuning(self.handle, words.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                               n.value, ctypes.byref(n)), "rn_model_export_tuning: not tuned")
        return words

    def import_tuning(self, words: np.ndarray) -> None:
        """Take over another model's tile table; refused (RnError) when it was measured for another
        architecture, element type, fusion setting or build.  Tiles change speed only."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        L.check(L.lib().rn_model_import_tuning(self.handle, words.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                               words.size), "rn_model_import_tuning: table of another model or build")

    def set_pair_fusion(self, on: bool) -> None:
        """Fused mode: conv3 + downsample of a stage's first block as one contraction (default
        on) or as two launches with the downsample tensor as the residual."""
        L.check(L.lib().rn_model_set_pair_fusion(self.handle, int(on)), "rn_model_set_pair_fusion")

    def set_streams(self, streams: int) -> None:
        """2 (default): batches of >= 128 images (fp32 at the default: >= 256) run as two halves on two
        streams; 1: one stream."""
        L.check(L.lib().rn_model_set_streams(self.handle, int(streams)), "rn_model_set_streams")

    def streams(self) -> int:
        return int(L.lib().rn_model_get_streams(self.handle))

    def parts(self, B: int) -> int:
        """Streams a forward of B images actually runs on: the configured count, halved while a part
        would be smaller than 64 images (128 for an fp32 model left at the library default)."""
        return int(L.lib().rn_model_parts(self.handle, int(B)))

    def set_chain(self, on: bool) -> None:
        """Fused mode: conv3 of a bottleneck block + conv1 of the next block as one launch wherever
        a chain kernel exists (fp32: the 64-channel blocks of stage 1; bf16: those and the
        128-channel blocks of stage 2; rn_model.c chain_applies).  Default on.  The same bits
        whatever the setting."""
        L.check(L.lib().rn_model_set_chain(self.handle, int(on)), "rn_model_set_chain")

    def set_stem_pool_fusion(self, on) -> None:
        """Fused mode: conv1 + bn1 + relu + maxpool as one launch (default on); 2 = that launch
        reads the NCHW input itself, no layout launch in front of it."""
        L.check(L.lib().rn_model_set_stem_pool_fusion(self.handle, int(on)), "rn_model_set_stem_pool_fusion")

    def set_front_parts(self, parts: int) -> None:
        """Stem, max-pool and first stage in `parts` slices of the batch (Infinity-Cache reuse)."""
        L.check(L.lib().rn_model_set_front_parts(self.handle, int(parts)), "rn_model_set_front_parts")

    def set_stem_exact(self, on: bool) -> None:
        """fp32: stem in the exact-K form (K = 160, default) or the 4-channel slot form (224)."""
        L.check(L.lib().rn_model_set_stem_exact(self.handle, int(on)), "rn_model_set_stem_exact")

    def set_profiling(self, on: bool) -> None:
        L.check(L.lib().rn_model_set_profiling(self.handle, int(on)), "rn_model_set_profiling")

    def profile(self) -> List[dict]:
        """Per-op records of the last profiled forward (after a sync)."""
        lib = L.lib()
        self.ctx.sync()
        recs = []
        op, layer = ctypes.c_char_p(), ctypes.c_char_p()
        ms, fl, by = ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
        for i in range(lib.rn_model_profile_count(self.handle)):
            L.check(lib.rn_model_profile_get(self.handle, i, ctypes.byref(op), ctypes.byref(layer),
                                             ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)),
                    "rn_model_profile_get")
            recs.append({"op": op.value.decode(), "layer": layer.value.decode(), "ms": ms.value,
                         "flops": fl.value, "bytes": by.value})
        return recs

    def activation_bytes(self) -> int:
        return int(L.lib().rn_model_activation_bytes(self.handle))

    def close(self) -> None:
        if self.handle:
            # refused (nothing freed) while a graph captured from this model lives: close those first
            L.check(L.lib().rn_model_destroy(self.handle), "rn_model_destroy: graphs captured from the model "
                    "are still alive")
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedModel:
    """One batch over several devices of a node (rn_shard_*): the multi-device form of the
    reference's main() (main.cu:228-254).  Contiguous batch split, weights replicated, no data
    moves between devices; one host thread + context + model per listed device inside the
    library.  A device may be listed twice (how a one-GPU box exercises the sharding)."""

    def __init__(self, devices, arch: str = "resnet50", state: Optional[Dict[str, np.ndarray]] = None,
                 weights_dir: Optional[str] = None, dtype: str = "f32"):
        lib = L.lib()
        if (state is None) == (weights_dir is None):
            raise ValueError("give exactly one of state / weights_dir")
        dev = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        L.check(lib.rn_shard_create(ctypes.byref(h), dev, len(devices), ARCH_ID[arch]), "rn_shard_create")
        self.handle, self.n = h, len(devices)
        self._stream_B = 0
        if weights_dir is not None:
            self._check(lib.rn_shard_load_dir(h, weights_dir.encode()), "rn_shard_load_dir")
        else:
            for key, arr in state.items():
                if key.endswith("num_batches_tracked"):
                    continue
                a = np.ascontiguousarray(arr, dtype=np.float32)
                self._check(lib.rn_shard_set_tensor(h, key.encode(), a.ctypes.data, a.size),
                            f"rn_shard_set_tensor({key})")
        self._check(lib.rn_shard_set_dtype(h, {"f32": L.RN_DTYPE_F32, "bf16": L.RN_DTYPE_BF16}[dtype]),
                    "rn_shard_set_dtype")
        self._check(lib.rn_shard_finalize(h), "rn_shard_finalize")

    def _check(self, status: int, where: str) -> None:
        if status != L.RN_OK:
            msg = L.lib().rn_shard_last_error(self.handle)
            raise L.RnError(status, where, msg.decode() if msg else "")

    @staticmethod
    def bounds(B: int, rank: int, world: int) -> Tuple[int, int]:
        lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
        L.lib().rn_shard_bounds(B, rank, world, ctypes.byref(lo), ctypes.byref(hi))
        return lo.value, hi.value

    def forward(self, x: np.ndarray, fused: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """NCHW host array -> (logits [B,1000], top-1 [B]) host arrays, image order."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        B = x.shape[0]
        logits = np.empty((B, 1000), dtype=np.float32)
        top1 = np.empty(B, dtype=np.uint64)
        self._check(L.lib().rn_shard_forward(self.handle, x.ctypes.data, B, logits.ctypes.data,
                                             top1.ctypes.data,
                                             L.RN_FWD_FUSED if fused else L.RN_FWD_REFERENCE_OPS),
                    "rn_shard_forward")
        return logits, top1

    def tune(self, x: np.ndarray, fused: bool = True) -> None:
        x = np.ascontiguousarray(x, dtype=np.float32)
        self._check(L.lib().rn_shard_tune(self.handle, x.ctypes.data, x.shape[0],
                                          L.RN_FWD_FUSED if fused else L.RN_FWD_REFERENCE_OPS),
                    "rn_shard_tune")

    def tuning_of(self, rank: int) -> np.ndarray:
        """The tile table shard `rank` runs with (rn_shard_model + rn_model_export_tuning)."""
        lib = L.lib()
        mh = lib.rn_shard_model(self.handle, rank)
        n = ctypes.c_uint64()
        self._check(lib.rn_model_export_tuning(mh, None, 0, ctypes.byref(n)), "rn_model_export_tuning")
        words = np.zeros(n.value, dtype=np.uint64)
        self._check(lib.rn_model_export_tuning(mh, words.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n.value,
                                               ctypes.byref(n)), "rn_model_export_tuning: shard not tuned")
        return words

    def placement(self, rank: int):
        """(device, NUMA node of the device's slot or -1, the CPUs shard `rank`'s host thread is bound to or '')."""
        dev, node, buf = ctypes.c_int(), ctypes.c_int(), ctypes.create_string_buffer(256)
        self._check(L.lib().rn_shard_placement(self.handle, rank, ctypes.byref(dev), ctypes.byref(node), buf, 256),
                    "rn_shard_placement")
        return dev.value, node.value, buf.value.decode()

    # streaming form: consecutive batches of B images, two in flight on every device
    def stream_open(self, B: int, fused: bool = True) -> None:
        self._check(L.lib().rn_shard_stream_open(self.handle, B, L.RN_FWD_FUSED if fused else L.RN_FWD_REFERENCE_OPS),
                    "rn_shard_stream_open")
        self._stream_B = B

    def stream_buffer(self, rank: int):
        """(pinned staging of shard `rank` for the next submit as [hi-lo,3,224,224], lo, hi)."""
        ptr, lo, hi = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64()
        self._check(L.lib().rn_shard_stream_buffer(self.handle, rank, ctypes.byref(ptr), ctypes.byref(lo),
                                                   ctypes.byref(hi)), "rn_shard_stream_buffer")
        n = hi.value - lo.value
        if n == 0:
            return None, lo.value, hi.value
        buf = (ctypes.c_float * (n * 3 * 224 * 224)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.float32).reshape(n, 3, 224, 224), lo.value, hi.value

    def submit(self, x: Optional[np.ndarray] = None) -> None:
        ptr = None
        if x is not None:
            x = np.ascontiguousarray(x, dtype=np.float32)
            assert self._stream_B == 0 or x.shape == (self._stream_B, 3, 224, 224)
            ptr = x.ctypes.data
        self._check(L.lib().rn_shard_submit(self.handle, ptr), "rn_shard_submit")

    def collect(self) -> Tuple[np.ndarray, np.ndarray]:
        logits = np.empty((self._stream_B, 1000), dtype=np.float32)
        top1 = np.empty(self._stream_B, dtype=np.uint64)
        self._check(L.lib().rn_shard_collect(self.handle, logits.ctypes.data, top1.ctypes.data), "rn_shard_collect")
        return logits, top1

    def in_flight(self) -> int:
        return int(L.lib().rn_shard_in_flight(self.handle))

    def stream_close(self) -> None:
        self._check(L.lib().rn_shard_stream_close(self.handle), "rn_shard_stream_close")

    def close(self) -> None:
        if self.handle:
            L.lib().rn_shard_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Graph:
    """A captured forward (rn_model_capture): launch() replays it on the context's stream."""

    def __init__(self, model: NativeModel, input_ptr, batch: int, logits_ptr, fused: bool = True):
        self.model = model
        h = ctypes.c_void_p()
        L.check(L.lib().rn_model_capture(model.handle, input_ptr, batch, logits_ptr,
                                         L.RN_FWD_FUSED if fused else L.RN_FWD_REFERENCE_OPS,
                                         ctypes.byref(h)), "rn_model_capture", model.ctx.handle)
        self.handle = h

    def launch(self) -> None:
        L.check(L.lib().rn_graph_launch(self.handle), "rn_graph_launch", self.model.ctx.handle)

    def node_count(self) -> int:
        return int(L.lib().rn_graph_node_count(self.handle))

    def close(self) -> None:
        if self.handle:
            L.lib().rn_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipeline:
    """Stream of host batches through a NativeModel with the upload of the next batch
    overlapped with the forward of the current one (rn_pipeline_*, two slots)."""

    def __init__(self, model: NativeModel, batch: int, fused: bool = True):
        self.model, self.batch = model, batch
        h = ctypes.c_void_p()
        L.check(L.lib().rn_pipeline_create(model.handle, ctypes.byref(h), batch,
                                           L.RN_FWD_FUSED if fused else L.RN_FWD_REFERENCE_OPS),
                "rn_pipeline_create", model.ctx.handle)
        self.handle = h

    def input_buffer(self) -> np.ndarray:
        """The pinned staging buffer of the next slot as a [B,3,224,224] array: fill it in
        place, then submit() without an argument."""
        ptr = ctypes.c_void_p()
        L.check(L.lib().rn_pipeline_input_buffer(self.handle, ctypes.byref(ptr)),
                "rn_pipeline_input_buffer", self.model.ctx.handle)
        n = self.batch * 3 * 224 * 224
        buf = (ctypes.c_float * n).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.float32).reshape(self.batch, 3, 224, 224)

    def submit(self, x: np.ndarray | None = None) -> None:
        """x: [n,3,224,224] with n <= batch (a ragged last batch), or None when the staging
        buffer was filled in place (a whole batch)."""
        ptr, n = None, self.batch
        if x is not None:
            x = np.ascontiguousarray(x, dtype=np.float32)
            assert x.shape[1:] == (3, 224, 224)
            ptr, n = x.ctypes.data, x.shape[0]
        L.check(L.lib().rn_pipeline_submit_n(self.handle, ptr, n), "rn_pipeline_submit_n",
                self.model.ctx.handle)

    def collect(self) -> np.ndarray:
        return self.collect_top1()[0]

    def collect_top1(self):
        """(logits [n,1000], class indices [n]) of the oldest batch in flight."""
        out = np.empty((self.batch, 1000), dtype=np.float32)
        idx = np.empty(self.batch, dtype=np.uint64)
        n = ctypes.c_uint64()
        L.check(L.lib().rn_pipeline_collect_n(self.handle, out.ctypes.data, idx.ctypes.data, ctypes.byref(n)),
                "rn_pipeline_collect_n", self.model.ctx.handle)
        return out[:n.value], idx[:n.value]

    def in_flight(self) -> int:
        return int(L.lib().rn_pipeline_in_flight(self.handle))

    def run(self, batches):
        """Yield logits for an iterable of host batches, keeping two in flight."""
        for x in batches:
            if self.in_flight() == 2:
                yield self.collect()
            self.submit(x)
        while self.in_flight():
            yield self.collect()

    def close(self) -> None:
        if self.handle:
            L.lib().rn_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
