#!/usr/bin/env python3
"""Phase breakdown of the contraction kernel's blocks from in-kernel wall-clock stamps.

    python tools/conv_stamps.py B H W Cin Cout k stride pad --cand 1 [--dtype bf16] [--residual]
Stamps (100 MHz): 0 block start, 1 first operands staged, 2 K loop done, 3 epilogue stores
issued, 4 stores acknowledged.  Prints medians in microseconds over all blocks.
--raw: every slot relative to slot 0 (conv_strip_kernel, candidate 15: 0 start, 1 first barrier
passed, 2-6 step 4: top / own memory landed / barrier passed / taps done / results packed, 7 end of
step 5, 10 last step done, 11 stores acknowledged)."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import resnet_c_amd as R
from resnet_c_amd import _lib as L
from resnet_c_amd.tensor import _DeviceBuffer

ap = argparse.ArgumentParser()
ap.add_argument("dims", type=int, nargs=8)
ap.add_argument("--cand", type=int, default=1)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--residual", action="store_true")
ap.add_argument("--exact", action="store_true", help="small-Cin exact-K form (fp32): dims are the unpadded image")
ap.add_argument("--raw", action="store_true", help="median of every stamp slot relative to slot 0")
ap.add_argument("--soak", type=float, default=1.5, help="seconds of back-to-back launches before the stamped one")
a = ap.parse_args()
B, H, W, Cin, Cout, k, s, p = a.dims
lib, ctx = L.lib(), R.get_ctx()
dt = L.RN_DTYPE_BF16 if a.dtype == "bf16" else L.RN_DTYPE_F32
es = 2 if a.dtype == "bf16" else 4
ho, wo = int(lib.rn_conv_output_size(H, k, s, p)), int(lib.rn_conv_output_size(W, k, s, p))
rng = np.random.default_rng(0)
def buf(n):
    b = _DeviceBuffer(ctx, n * es)
    h = rng.standard_normal(n, dtype=np.float32) * 0.5
    if es == 2: h = R.ops.to_bf16_bits(h)
    L.check(lib.rn_memcpy_h2d(ctx.handle, b.ptr, h.ctypes.data, h.nbytes), "h2d", ctx.handle)
    return b
if a.exact:
    Hp, Wp = H + 2 * p, W + 2 * p
    x = buf(B * Hp * Wp * Cin); wn = int(lib.rn_conv2d_packed_weight_numel_exact(Cin, Cout, k)); w = buf(wn)
else:
    x = buf(B * H * W * max(Cin, 4) if Cin < 4 else B * H * W * Cin); wn = int(lib.rn_conv2d_packed_weight_numel_dt(dt, Cin, Cout, k)); w = buf(wn)
out = _DeviceBuffer(ctx, B * ho * wo * Cout * es); res = buf(B * ho * wo * Cout) if a.residual else None
sc = R.FloatTensor.from_numpy(np.ones(Cout, np.float32), R.Device.GPU)
ep = L.Epilogue(sc.data(), sc.data(), res.ptr if res else None, 1)
nblk = 1 << 16
st = _DeviceBuffer(ctx, nblk * 128)
lib.rn_ctx_set_conv_tile(ctx.handle, a.cand)
def run():
    if a.exact:
        L.check(lib.rn_conv2d_nhwc_exact_forward(ctx.handle, x.ptr, out.ptr, w.ptr, k, s, ho, wo, B, Cin, Cout, Hp, Wp, ctypes.byref(ep)), "conv", ctx.handle)
        return
    L.check(lib.rn_conv2d_nhwc_forward_dt(ctx.handle, dt, dt, x.ptr, out.ptr, w.ptr, k, s, p, ho, wo, B, Cin, Cout, H, W, ctypes.byref(ep)), "conv", ctx.handle)
run(); run(); ctx.sync()
import time
t_end = time.time() + a.soak
while time.time() < t_end:
    for _ in range(50): run()
    ctx.sync()
lib.rn_memset(ctx.handle, st.ptr, 0, nblk * 128)
lib.rn_ctx_set_debug_stamps(ctx.handle, st.ptr)
run(); ctx.sync()
lib.rn_ctx_set_debug_stamps(ctx.handle, None)
raw = np.empty(nblk * 16, dtype=np.uint64)
lib.rn_memcpy_d2h(ctx.handle, raw.ctypes.data, st.ptr, raw.nbytes)
full = raw.reshape(nblk, 16).astype(np.float64)
if a.raw:
    full = full[full[:, 0] > 0]
    print(f"blocks {len(full)}  span {(full.max() - full[:, 0].min()) / 100:.1f} us")
    for i in range(16):
        col = full[:, i]
        ok = col > 0
        if ok.any():
            rel = (col[ok] - full[ok, 0]) / 100
            print(f"  slot {i:2d}: median {np.median(rel):8.2f} us   p10 {np.percentile(rel, 10):8.2f}   p90 {np.percentile(rel, 90):8.2f}   blocks {ok.sum()}")
    sys.exit(0)
t = full[:, :5]
keep = t[:, 0] > 0
full = full[keep]
t = t[keep]
us = t / 100.0
d = np.diff(us, axis=1)
print(f"blocks {len(t)}  kernel span {(us[:,4].max()-us[:,0].min()):.1f} us")
for i, name in enumerate(["start->operands staged", "K loop (+next-tile prefetch)", "epilogue until stores issued", "stores acknowledged"]):
    print(f"  {name:32s} median {np.median(d[:, i]):7.2f} us   p90 {np.percentile(d[:, i], 90):7.2f}")
print(f"  block lifetime                   median {np.median(us[:,4]-us[:,0]):7.2f} us")

if full[:, 5].max() > 0:
    print(f"  epilogue: K-loop end -> acc in LDS issued   median {np.median(full[:,5]-full[:,2])/100:7.2f} us")
    print(f"  epilogue: barrier                           median {np.median(full[:,6]-full[:,5])/100:7.2f} us")
    print(f"  epilogue: read back + stores issued         median {np.median(full[:,3]-full[:,6])/100:7.2f} us")
if full[:, 7].max() > 0:
    print(f"  prologue: start -> set-up done              median {np.median(full[:,7]-full[:,0])/100:7.2f} us")
    print(f"  prologue: loads issued -> staged + barrier  median {np.median(full[:,1]-full[:,7])/100:7.2f} us")
if full[:, 9].max() > 0:
    ghz = (full[:, 9] - full[:, 8]) / np.maximum(full[:, 2] - full[:, 1], 1) * 0.1
    print(f"  shader clock inside the K loop              median {np.median(ghz):7.3f} GHz   p10 {np.percentile(ghz,10):.3f}  p90 {np.percentile(ghz,90):.3f}")
