#!/usr/bin/env python3
"""Where a steady-state item of the fused stem + max-pool kernel spends its time: shader-clock
stamps (s_memtime) of every wave of every block around the phases of the block's fourth item.

    python tools/stem_stamps.py [--dtype bf16|f32] [--batch 256]
Slots: 0 item top, 1 next patch's DMA pieces issued, 3 this wave's ship reached (three kernel rows
into tile 0 for waves 0-3, into tile 1 for waves 4-7), 4 its pooled row shipped, 5 first tile's
chain done, 6 last tile's chain done, 7 last epilogue done, 8 barrier passed.  Prints median cycles
between consecutive slots per wave row (wm = wave >> 1: rows 0, 1 multiply four tiles, rows 2, 3
three at 224 x 224)."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import resnet_c_amd as R
from resnet_c_amd import _lib as L
from resnet_c_amd.tensor import _DeviceBuffer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
B = a.batch
lib, ctx = L.lib(), R.get_ctx()
x = np.concatenate([R.weights.generate_input(32, seed=5)] * ((B + 31) // 32))[:B]
w = R.weights.generate_tensor("conv1.weight", (64, 3, 7, 7), 0)
xin = R.FloatTensor.from_numpy(x, R.Device.GPU)
wd = R.FloatTensor.from_numpy(w, R.Device.GPU)
sc = R.FloatTensor.from_numpy(np.full(64, 1.1, np.float32), R.Device.GPU)
dt, es, cpad = (L.RN_DTYPE_BF16, 2, 4) if a.dtype == "bf16" else (L.RN_DTYPE_F32, 4, 3)
xp = _DeviceBuffer(ctx, B * 230 * 230 * cpad * es)
L.check(lib.rn_nchw_to_nhwc_pad_dt(ctx.handle, dt, xin.data(), xp.ptr, B, 3, 224, 224, cpad, 3), "pad", ctx.handle)
wp = _DeviceBuffer(ctx, int(lib.rn_stem_pool_packed_weight_numel(dt)) * es)
L.check(lib.rn_stem_pool_pack_weight_dt(ctx.handle, dt, wd.data(), wp.ptr, 3), "pack", ctx.handle)
out = _DeviceBuffer(ctx, B * 56 * 56 * 64 * es)

def run():
    L.check(lib.rn_stem_pool_forward_dt(ctx.handle, dt, xp.ptr, out.ptr, wp.ptr, sc.data(), sc.data(), 1, B, 230, 230),
            "stem", ctx.handle)

t_end = time.time() + 1.0
while time.time() < t_end:
    for _ in range(50):
        run()
    ctx.sync()
nblk = 4096
st = _DeviceBuffer(ctx, nblk * 8 * 16 * 8)
lib.rn_memset(ctx.handle, st.ptr, 0, nblk * 8 * 16 * 8)
lib.rn_ctx_set_debug_stamps(ctx.handle, st.ptr)
run()
ctx.sync()
lib.rn_ctx_set_debug_stamps(ctx.handle, None)
h = np.empty(nblk * 8 * 16, dtype=np.uint64)
L.check(lib.rn_memcpy_d2h(ctx.handle, h.ctypes.data, st.ptr, h.nbytes), "d2h", ctx.handle)
h = h.reshape(nblk, 8, 16)
h = h[h[:, 0, 0] != 0]
print(f"{a.dtype} B={B}: {h.shape[0]} blocks stamped; median cycles between slots, by wave row")
pairs = [(0, 1, "DMA issue"), (3, 4, "ship"), (0, 5, "top -> tile 0 done"), (5, 6, "tiles 1.."), (6, 7, "last epilogue"),
         (7, 8, "barrier wait"), (0, 8, "item")]
for wave in range(8):
    d = h[:, wave, :9].astype(np.int64)
    print(f"wave {wave} (row {wave >> 1}): " + "  ".join(f"{n} {int(np.median(d[:, b] - d[:, a]))}" for a, b, n in pairs))
