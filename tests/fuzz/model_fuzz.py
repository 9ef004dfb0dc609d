#!/usr/bin/env python3
"""Randomised state-machine run of the model driver: forwards of random batch sizes interleaved with
every switch that only reschedules the same arithmetic (streams, depth-first front, chains, tuning,
graph capture and replay, the host pipeline; NOT pair fusion or stem fusion, which fold the batch-norm
scales into the weights / use another K padding and so round differently).  None of them may change a bit: every image's logits must
equal the ones a plain forward of the whole pool gave at the start (batch invariance), whatever ran
before -- arenas resized, scratch regrown, tiles tuned at another batch size.

    python tests/fuzz/model_fuzz.py [--seconds 60] [--seed 0] [--dtype f32|bf16] [--arch resnet50]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import resnet_c_amd as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--arch", default="resnet50")
    a = ap.parse_args()
    g = np.random.default_rng(a.seed)
    state = R.weights.generate_state(a.arch, 0)
    m = R.NativeModel(a.arch, state=state, dtype=a.dtype)
    pool = R.weights.generate_input(320, seed=1000 + a.seed)
    m.set_streams(1)
    base = m.forward(pool, fused=True)            # the reference bits of every image of the pool
    assert np.isfinite(base).all()
    if a.dtype == "f32":
        assert np.array_equal(m.forward(pool[:7], fused=False)[:7].argmax(1), base[:7].argmax(1))
    counts = {}
    t0 = time.time()
    try:
        while time.time() - t0 < a.seconds:
            op = str(g.choice(["forward"] * 6 + ["streams", "front", "chain", "tune", "graph", "pipeline"]))
            counts[op] = counts.get(op, 0) + 1
            if op == "forward":
                B = int(g.choice([1, 2, 3, 5, 8, 17, 31, 64, 65, 127, 128, 129, 200, 256, 300]))
                lo = int(g.integers(0, 320 - B + 1))
                got = m.forward(pool[lo:lo + B], fused=True)
                assert np.array_equal(got, base[lo:lo + B]), f"forward B={B} at {lo} after {counts}"
            elif op == "streams":
                m.set_streams(int(g.choice([1, 2, 4])))
            elif op == "front":
                m.set_front_parts(int(g.choice([1, 2, 4, 8])))
            elif op == "chain":
                m.set_chain(int(g.choice([0, 1])))
            elif op == "tune":
                B = int(g.choice([4, 32, 100, 128, 256]))
                xin = R.FloatTensor.from_numpy(pool[:B], R.Device.GPU)
                out = R.FloatTensor((B, 1000), R.Device.GPU)
                m.tune(xin.data(), B, out.data(), True)
                m.ctx.sync()
                assert np.array_equal(out.numpy(), base[:B]), f"tune B={B} after {counts}"
            elif op == "graph":
                B = int(g.choice([1, 6, 64, 130]))
                lo = int(g.integers(0, 320 - B + 1))
                xin = R.FloatTensor.from_numpy(pool[lo:lo + B], R.Device.GPU)
                out = R.FloatTensor((B, 1000), R.Device.GPU)
                gr = R.Graph(m, xin.data(), B, out.data(), True)
                R._lib.check(R._lib.lib().rn_memset(m.ctx.handle, out.data(), 0, B * 4000), "memset", m.ctx.handle)
                gr.launch(); gr.launch(); m.ctx.sync()
                assert np.array_equal(out.numpy(), base[lo:lo + B]), f"graph B={B} after {counts}"
                gr.close()
            else:
                B = int(g.choice([3, 40, 128]))
                pipe = R.Pipeline(m, B, fused=True)
                los = [int(g.integers(0, 320 - B + 1)) for _ in range(3)]
                outs = list(pipe.run([pool[lo:lo + B] for lo in los]))
                for lo, o in zip(los, outs):
                    assert np.array_equal(o, base[lo:lo + B]), f"pipeline B={B} after {counts}"
                pipe.close()
    finally:
        m.close()
    print(f"model_fuzz {a.arch} {a.dtype}: {counts}, seed {a.seed}: every forward, tuned forward, graph replay and "
          f"pipelined batch gave the pool's bits")


if __name__ == "__main__":
    main()
