#!/usr/bin/env python3
"""Small-batch forward of two builds of librn_hip.so side by side, layer by layer (plain ctypes on the
entry points both builds have: the model driver and its per-op profile).

    python tools/ab_b1.py .ab/librn_hip_r2.so resnet.c_amd/librn_hip.so [--batch 1]

Each library is loaded in a process of its own (two HIP runtimes' worth of symbols do not mix)."""
import argparse, ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(path, B):
    import numpy as np
    try:
        import torch  # noqa: F401  (same HIP runtime instance as the package uses)
    except Exception:
        pass
    from resnet_c_amd import weights
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    vp, u64 = ctypes.c_void_p, ctypes.c_uint64
    ctx, m = vp(), vp()
    lib.rn_ctx_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, vp]
    assert lib.rn_ctx_create(ctypes.byref(ctx), 0, None) == 0
    lib.rn_model_create.argtypes = [vp, ctypes.POINTER(vp), ctypes.c_int]
    assert lib.rn_model_create(ctx, ctypes.byref(m), 50) == 0
    lib.rn_model_set_tensor.argtypes = [vp, ctypes.c_char_p, vp, u64]
    for k, v in weights.generate_state("resnet50", 0).items():
        if k.endswith("num_batches_tracked"):
            continue
        a = np.ascontiguousarray(v, dtype=np.float32)
        assert lib.rn_model_set_tensor(m, k.encode(), a.ctypes.data, a.size) == 0, k
    lib.rn_model_finalize.argtypes = [vp]
    assert lib.rn_model_finalize(m) == 0
    x = weights.generate_input(B, 0)
    dx, dl = vp(), vp()
    lib.rn_malloc.argtypes = [vp, ctypes.POINTER(vp), u64]
    lib.rn_malloc(ctx, ctypes.byref(dx), x.nbytes); lib.rn_malloc(ctx, ctypes.byref(dl), B * 4000)
    lib.rn_memcpy_h2d.argtypes = [vp, vp, vp, u64]
    lib.rn_memcpy_h2d(ctx, dx, x.ctypes.data, x.nbytes)
    for f in (lib.rn_model_forward, lib.rn_model_tune):
        f.argtypes = [vp, vp, u64, vp, ctypes.c_int]
    lib.rn_sync.argtypes = [vp]
    assert lib.rn_model_tune(m, dx, B, dl, 1) == 0
    import time
    for _ in range(20):
        lib.rn_model_forward(m, dx, B, dl, 1)
    lib.rn_sync(ctx)
    t0 = time.perf_counter()
    for _ in range(300):
        lib.rn_model_forward(m, dx, B, dl, 1)
    lib.rn_sync(ctx)
    print(f"forward B={B}: {(time.perf_counter() - t0) / 300 * 1e6:8.1f} us back to back")
    lib.rn_model_set_profiling.argtypes = [vp, ctypes.c_int]
    lib.rn_model_profile_count.argtypes = [vp]; lib.rn_model_profile_count.restype = u64
    lib.rn_model_profile_get.argtypes = [vp, u64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p),
                                         ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_double)]
    lib.rn_model_set_profiling(m, 1)
    best = {}
    order = []
    for _ in range(5):
        lib.rn_model_forward(m, dx, B, dl, 1)
        for i in range(lib.rn_model_profile_count(m)):
            op, layer, ms, fl, by = ctypes.c_char_p(), ctypes.c_char_p(), ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
            lib.rn_model_profile_get(m, i, ctypes.byref(op), ctypes.byref(layer), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by))
            k = layer.value.decode()
            if k not in best:
                order.append(k)
            best[k] = min(best.get(k, 1e9), ms.value * 1e3)
    for k in order:
        print(f"LAYER {k:44s} {best[k]:8.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs=2)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    tables = []
    for lib in a.libs:
        r = subprocess.run([sys.executable, __file__, "--child", os.path.abspath(lib), str(a.batch)], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:]); sys.exit(1)
        t = {}
        for line in r.stdout.splitlines():
            if line.startswith("LAYER"):
                p = line.split()
                t[p[1]] = float(p[2])
            else:
                print(os.path.basename(lib), line)
        tables.append(t)
    names = list(tables[1]) + [k for k in tables[0] if k not in tables[1]]
    print(f"{'layer (us by events, min of 5)':46s} {os.path.basename(a.libs[0])[:18]:>18s} {os.path.basename(a.libs[1])[:18]:>18s} {'delta':>8s}")
    for k in names:
        x, y = tables[0].get(k), tables[1].get(k)
        d = f"{y - x:8.1f}" if x is not None and y is not None else ""
        print(f"{k:46s} {x if x is not None else float('nan'):18.1f} {y if y is not None else float('nan'):18.1f} {d}")
    print(f"{'sum':46s} {sum(tables[0].values()):18.1f} {sum(tables[1].values()):18.1f} {sum(tables[1].values()) - sum(tables[0].values()):8.1f}")
