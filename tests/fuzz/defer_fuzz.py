#!/usr/bin/env python3
"""Randomised programs of the seven reference ops on a DEFERRED context against the same programs run
literally (rn_ctx_set_deferred, rn_defer.hip): test infrastructure under tests/.

    python tests/fuzz/defer_fuzz.py [--seconds 60] [--seed 0]

A program is a random sequence of conv / batch-norm / ReLU / add / max-pool / avg-pool calls over a small pool
of device tensors -- in place and out of place, chains that fold (conv -> bn -> add -> relu on one buffer) and
chains that do not (another order, another buffer, a shape without an NHWC contraction), residuals that are
NHWC-tagged outputs of earlier convolutions or plain NCHW tensors -- interleaved with the things that observe
or disturb recorded work: whole and partial reads, rn_observe, rn_flush, writes into operands, frees, parameter
updates.  After every read and at the end, every live tensor must hold what the literal run holds: bit for bit
where no batch-norm was folded into a convolution, within the fused epilogue's rounding otherwise (the literal
route applies ops.cu:150's double expression, the fused one a folded fp32 fmaf)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import resnet_c_amd as R
from resnet_c_amd import _lib as L


def call(ctx, name, *args):
    L.check(getattr(L.lib(), name)(ctx.handle, *args), name, ctx.handle)


class Program:
    """A recorded program: a list of (kind, args) over tensor slots; replayable on either kind of context."""

    def __init__(self, g):
        self.g = g
        B = int(g.integers(1, 4))
        self.B = B
        # a few shapes that chain: (C, H, W) with C from the contraction's and the direct kernel's families
        C0 = int(g.choice([32, 64, 96, 5, 3]))
        H = int(g.integers(4, 13))
        W = int(g.integers(4, 13))
        stem = g.random() < 0.12
        if stem:   # the reference's first four ops on an image the fused stem launch takes (or just not: W = 18, 20)
            C0, H, W = int(g.choice([3, 3, 1])), int(g.integers(7, 24)), int(g.choice([16, 32, 32, 18, 20]))
        self.shapes = {0: (B, C0, H, W)}
        self.host = {0: g.standard_normal((B, C0, H, W), dtype=np.float32)}
        self.steps = []
        self.params = {}
        nxt = 1
        live = [0]
        if stem:
            # conv 7x7 / 2 / 3 -> bn -> relu in place, max-pool 3x3 / 2 / 1 into another tensor (main.cu:179-192):
            # one launch that writes both tensors; sometimes a link is missing or something reads in between
            ho, wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
            self.shapes[1] = (B, 64, ho, wo)
            w = (g.standard_normal((64, C0, 7, 7), dtype=np.float32) / np.sqrt(C0 * 49)).astype(np.float32)
            self.steps.append(("conv", 0, 1, 0, 7, 2, 3))
            self.params[0] = [w]
            live.append(1)
            if g.random() < 0.85:
                self._bn(1, 1)
            if g.random() < 0.9:
                self.steps.append(("relu", 1, 1))
            if g.random() < 0.2:
                self.steps.append((str(g.choice(["observe", "read", "partial", "flush"])), int(g.choice([0, 1]))))
            self.shapes[2] = (B, 64, (ho + 2 - 3) // 2 + 1, (wo + 2 - 3) // 2 + 1)
            self.steps.append(("maxpool", 1, 2, 3, 2, 1))
            live.append(2)
            nxt = 3
        n_ops = int(g.integers(4, 14))
        for _ in range(n_ops):
            kind = str(g.choice(["conv", "conv", "conv", "bn", "relu", "add", "maxpool", "avgpool", "observe", "read",
                                 "partial", "flush", "rewrite", "free", "newbn", "slice_relu", "slice_conv", "view_conv",
                                 "view_bn", "boundary"]))
            src = int(g.choice(live))
            Bs, C, Hs, Ws = self.shapes[src]
            if kind == "conv":
                k = int(g.choice([1, 3])) if C != 3 else int(g.choice([3, 7]))
                s = int(g.choice([1, 1, 2]))
                p = int(g.choice([0, k // 2]))
                if Hs + 2 * p < k or Ws + 2 * p < k:
                    continue
                Cout = int(g.choice([32, 64, 8, 100]))
                ho, wo = (Hs + 2 * p - k) // s + 1, (Ws + 2 * p - k) // s + 1
                dst = nxt
                nxt += 1
                self.shapes[dst] = (Bs, Cout, ho, wo)
                w = (g.standard_normal((Cout, C, k, k), dtype=np.float32) / np.sqrt(C * k * k)).astype(np.float32)
                self.steps.append(("conv", src, dst, len(self.params), k, s, p))
                self.params[len(self.params)] = [w]
                live.append(dst)
                # what follows a convolution in a network, in a random subset and sometimes in another order
                tail = [t for t in ("bn", "add", "relu") if g.random() < 0.7]
                if g.random() < 0.15:
                    tail = list(g.permutation(tail))
                for t in tail:
                    if t == "bn":
                        self._bn(dst, dst)
                    elif t == "relu":
                        self.steps.append(("relu", dst, dst))
                    else:
                        same = [i for i in live if i != dst and self.shapes[i] == self.shapes[dst]]
                        if same:
                            r = int(g.choice(same))
                            self.steps.append(("add", dst, r, dst) if g.random() < 0.8 else ("add", r, dst, dst))
            elif kind == "boundary" and C == 64:
                # a block boundary of layer1 (main.cu:131-164): conv3 -> bn -> add -> relu of a 64-channel block, conv1
                # -> bn -> relu of the next on its output -- one launch when nothing comes between (rn_chain.hip);
                # sometimes something does
                def conv1x1(a, Cin_, Cout_):
                    nonlocal nxt
                    d = nxt
                    nxt += 1
                    self.shapes[d] = (Bs, Cout_, Hs, Ws)
                    w = (g.standard_normal((Cout_, Cin_, 1, 1), dtype=np.float32) / np.sqrt(Cin_)).astype(np.float32)
                    self.steps.append(("conv", a, d, len(self.params), 1, 1, 0))
                    self.params[len(self.params)] = [w]
                    live.append(d)
                    return d
                r = conv1x1(src, 64, 256)
                y = conv1x1(src, 64, 256)
                self._bn(y, y)
                self.steps.append(("add", y, r, y) if g.random() < 0.8 else ("add", r, y, y))
                self.steps.append(("relu", y, y))
                if g.random() < 0.3:
                    self.steps.append((str(g.choice(["observe", "read", "partial", "flush"])), int(g.choice([y, r, src]))))
                if g.random() < 0.4:
                    # the projection shortcut of the next stage stands between (main.cu:131-137 runs it first); the
                    # chain launch moves conv1 in front of it -- unless this group touches conv1's buffers
                    d = conv1x1(y, 256, int(g.choice([32, 64])))
                    self._bn(d, d)
                    if g.random() < 0.3:
                        self.steps.append(("relu", d, d))
                    if g.random() < 0.15:
                        self.steps.append(("add", d, d, d))
                t1 = conv1x1(y, 256, int(g.choice([64, 128])))
                if g.random() < 0.8:
                    self._bn(t1, t1)
                if g.random() < 0.9:
                    self.steps.append(("relu", t1, t1))
            elif kind == "bn":
                dst = src if g.random() < 0.7 else self._new_like(src, live)
                nxt = max(nxt, dst + 1)
                self._bn(src, dst)
            elif kind == "relu":
                dst = src if g.random() < 0.7 else self._new_like(src, live)
                nxt = max(nxt, dst + 1)
                self.steps.append(("relu", src, dst))
            elif kind == "add":
                same = [i for i in live if self.shapes[i] == self.shapes[src]]
                other = int(g.choice(same))
                dst = src if g.random() < 0.7 else self._new_like(src, live)
                nxt = max(nxt, dst + 1)
                self.steps.append(("add", src, other, dst))
            elif kind in ("maxpool", "avgpool"):
                k = int(g.choice([2, 3]))
                s = int(g.choice([1, 2]))
                p = int(g.choice([0, 1])) if kind == "maxpool" and k == 3 else 0
                if Hs + 2 * p < k or Ws + 2 * p < k:
                    continue
                ho, wo = (Hs + 2 * p - k) // s + 1, (Ws + 2 * p - k) // s + 1
                dst = nxt
                nxt += 1
                self.shapes[dst] = (Bs, C, ho, wo)
                self.steps.append((kind, src, dst, k, s, p))
                live.append(dst)
            elif kind in ("observe", "read", "partial", "flush"):
                self.steps.append((kind, src))
            elif kind == "view_conv" and C % 64 == 0:
                # the same buffer under another shape (Tensor::view): [B, C, H, W] read as [B, C/2, H, 2W]
                dst = nxt
                nxt += 1
                self.shapes[dst] = (Bs, 32, Hs, 2 * Ws)
                w = (g.standard_normal((32, C // 2, 1, 1), dtype=np.float32) / np.sqrt(C // 2)).astype(np.float32)
                self.steps.append(("view_conv", src, dst, len(self.params)))
                self.params[len(self.params)] = [w]
                live.append(dst)
            elif kind == "view_bn" and C % 2 == 0:
                self.steps.append(("view_bn", src, len(self.params)))   # in place, as [B, C/2, H, 2W]
                C2 = C // 2
                self.params[len(self.params)] = [g.random(C2, dtype=np.float32) + 0.5, g.standard_normal(C2, dtype=np.float32) * 0.1,
                                                 g.standard_normal(C2, dtype=np.float32) * 0.1, g.random(C2, dtype=np.float32) + 0.5]
            elif kind == "slice_relu" and Bs > 1:
                # in place on ONE image of a batch: a pointer offset into what may be an NHWC-tagged buffer
                self.steps.append(("slice_relu", src, int(g.integers(0, Bs))))
            elif kind == "slice_conv" and Bs > 1 and C % 32 == 0:
                b0 = int(g.integers(0, Bs))
                dst = nxt
                nxt += 1
                self.shapes[dst] = (1, 32, Hs, Ws)
                w = (g.standard_normal((32, C, 1, 1), dtype=np.float32) / np.sqrt(C)).astype(np.float32)
                self.steps.append(("slice_conv", src, dst, len(self.params), b0))
                self.params[len(self.params)] = [w]
                live.append(dst)
            elif kind == "rewrite" and src != 0:
                self.steps.append(("rewrite", src, len(self.params)))
                self.params[len(self.params)] = [g.standard_normal(self.shapes[src], dtype=np.float32)]
            elif kind == "free" and src != 0 and len(live) > 2:
                live.remove(src)
                self.steps.append(("free", src))
            elif kind == "newbn":
                bns = [st for st in self.steps if st[0] == "bn"]
                if bns:
                    st = bns[int(g.integers(0, len(bns)))]
                    self.steps.append(("newbn", st[3], len(self.params)))
                    C_ = self.params[st[3]][0].shape[0]
                    self.params[len(self.params)] = [(g.random(C_, dtype=np.float32) + 0.5)]
        self.live_end = list(live)

    def _new_like(self, src, live):
        dst = max(self.shapes) + 1
        self.shapes[dst] = self.shapes[src]
        live.append(dst)
        return dst

    def _bn(self, src, dst):
        C = self.shapes[src][1]
        g = self.g
        pr = [g.random(C, dtype=np.float32) + 0.5, g.standard_normal(C, dtype=np.float32) * 0.1,
              g.standard_normal(C, dtype=np.float32) * 0.1, g.random(C, dtype=np.float32) + 0.5]
        self.steps.append(("bn", src, dst, len(self.params)))
        self.params[len(self.params)] = pr

    def run(self, ctx):
        """Execute on ctx; returns (reads in program order, final contents of the live tensors)."""
        T = {0: R.FloatTensor.from_numpy(self.host[0], R.Device.GPU)}
        P = {}
        reads = []

        def tensor(i):
            if i not in T:
                T[i] = R.FloatTensor(self.shapes[i], R.Device.GPU)
                # a defined starting content (literal and deferred runs must agree on bytes never written)
                call(ctx, "rn_memset", T[i].data(), 0, int(np.prod(self.shapes[i])) * 4)
            return T[i]

        def param(i):
            if i not in P:
                P[i] = [R.FloatTensor.from_numpy(a, R.Device.GPU) for a in self.params[i]]
            return P[i]

        for st in self.steps:
            kind = st[0]
            if kind == "conv":
                _, src, dst, pi, k, s, p = st
                Bs, C, Hs, Ws = self.shapes[src]
                _, Cout, ho, wo = self.shapes[dst]
                call(ctx, "rn_conv2d_forward", tensor(src).data(), tensor(dst).data(), param(pi)[0].data(), k, s, p,
                     ho, wo, Bs, C, Cout, Hs, Ws)
            elif kind == "bn":
                _, src, dst, pi = st
                Bs, C, Hs, Ws = self.shapes[src]
                call(ctx, "rn_batchnorm2d_forward", tensor(src).data(), tensor(dst).data(), *(q.data() for q in param(pi)),
                     Bs, C, Hs * Ws)
            elif kind == "relu":
                _, src, dst = st
                call(ctx, "rn_relu_forward", tensor(src).data(), tensor(dst).data(), int(np.prod(self.shapes[src])))
            elif kind == "add":
                _, a, b, dst = st
                call(ctx, "rn_add_forward", tensor(a).data(), tensor(b).data(), tensor(dst).data(),
                     int(np.prod(self.shapes[a])))
            elif kind in ("maxpool", "avgpool"):
                _, src, dst, k, s, p = st
                Bs, C, Hs, Ws = self.shapes[src]
                _, _, ho, wo = self.shapes[dst]
                call(ctx, "rn_maxpool2d_forward" if kind == "maxpool" else "rn_avgpool2d_forward", tensor(src).data(),
                     tensor(dst).data(), k, s, p, ho, wo, Bs, C, Hs, Ws)
            elif kind == "view_conv":
                _, src, dst, pi = st
                Bs, C, Hs, Ws = self.shapes[src]
                call(ctx, "rn_conv2d_forward", tensor(src).data(), tensor(dst).data(), param(pi)[0].data(), 1, 1, 0,
                     Hs, 2 * Ws, Bs, C // 2, 32, Hs, 2 * Ws)
            elif kind == "view_bn":
                _, src, pi = st
                Bs, C, Hs, Ws = self.shapes[src]
                call(ctx, "rn_batchnorm2d_forward", tensor(src).data(), tensor(src).data(), *(q.data() for q in param(pi)),
                     Bs, C // 2, 2 * Hs * Ws)
            elif kind == "slice_relu":
                _, src, b0 = st
                Bs, C, Hs, Ws = self.shapes[src]
                n = C * Hs * Ws
                call(ctx, "rn_relu_forward", tensor(src).data() + 4 * n * b0, tensor(src).data() + 4 * n * b0, n)
            elif kind == "slice_conv":
                _, src, dst, pi, b0 = st
                Bs, C, Hs, Ws = self.shapes[src]
                call(ctx, "rn_conv2d_forward", tensor(src).data() + 4 * C * Hs * Ws * b0, tensor(dst).data(),
                     param(pi)[0].data(), 1, 1, 0, Hs, Ws, 1, C, 32, Hs, Ws)
            elif kind == "observe":
                ctx.observe(tensor(st[1]).data())
            elif kind == "flush":
                ctx.flush()
            elif kind == "read":
                reads.append((st[1], tensor(st[1]).numpy()))
            elif kind == "partial":
                n = int(np.prod(self.shapes[st[1]]))
                lo = n // 3
                cnt = max(1, n // 2)
                cnt = min(cnt, n - lo)
                h = np.empty(cnt, dtype=np.float32)
                call(ctx, "rn_memcpy_d2h", h.ctypes.data, tensor(st[1]).data() + 4 * lo, 4 * cnt)
                reads.append((st[1], h))
            elif kind == "rewrite":
                a = self.params[st[2]][0]
                call(ctx, "rn_memcpy_h2d", tensor(st[1]).data(), a.ctypes.data, a.nbytes)
            elif kind == "free":
                T.pop(st[1], None)   # FloatTensor.__del__ -> rn_free
            elif kind == "newbn":
                a = self.params[st[2]][0]
                call(ctx, "rn_memcpy_h2d", param(st[1])[0].data(), a.ctypes.data, a.nbytes)
        final = {i: tensor(i).numpy() for i in self.live_end}
        return reads, final


def close(a, b, exact):
    if exact:
        return np.array_equal(a, b, equal_nan=True)
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    return bool(np.abs(a - b).max() <= 3e-5 * scale) if a.size else True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    g = np.random.default_rng(a.seed)
    ctx = R.get_ctx()
    t0 = time.time()
    n = fused = 0
    while time.time() - t0 < a.seconds:
        prog = Program(g)
        ctx.set_deferred(False)
        want_reads, want_final = prog.run(ctx)
        ctx.set_deferred(True)
        s0 = ctx.deferred_stats()
        got_reads, got_final = prog.run(ctx)
        s1 = ctx.deferred_stats()
        ctx.set_deferred(False)   # the barrier: nothing pending, nothing tagged afterwards
        s2 = ctx.deferred_stats()
        assert s2["pending_ops"] == 0 and s2["nhwc_buffers"] == 0, (a.seed, n, s2)
        # a fused launch that folded a batch-norm rounds differently from ops.cu:150's double expression; everything
        # else is the literal kernels on the same values
        folded_bn = any(st[0] == "bn" for st in prog.steps) and s1["fused_launches"] > s0["fused_launches"]
        # (and a 3-channel convolution runs in its exact-K form when recorded, in the 4-channel / 8-slot form
        # literally: the same products grouped into other K tiles)
        folded_bn = folded_bn or any(st[0] == "conv" and prog.shapes[st[1]][1] <= 4 for st in prog.steps)
        assert len(got_reads) == len(want_reads)
        for (i, x), (j, y) in zip(got_reads, want_reads):
            assert i == j and close(x, y, not folded_bn), f"seed {a.seed} program {n}: read of tensor {i} differs\n{prog.steps}"
        for i in want_final:
            assert close(got_final[i], want_final[i], not folded_bn), \
                f"seed {a.seed} program {n}: final tensor {i} {prog.shapes[i]} differs\n{prog.steps}"
        n += 1
        fused += s1["fused_launches"] - s0["fused_launches"]
    print(f"{n} random programs on a deferred context ({fused} fused launches), every read and every final tensor "
          f"as the literal run's: all within tolerance")


if __name__ == "__main__":
    main()
