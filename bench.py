#!/usr/bin/env python3
"""Throughput bench of the hot path: ResNet-50 fp32 forward, batch 256 per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward pass of the whole network over one batch of 256 synthetic
224x224x3 images that are already resident in HBM (BASELINE.json configs[2]).
With N GPUs every rank runs the same step on its own 256-image shard of a global
batch of N*256 (batch split, weights replicated, NO data-path collective -- the
forward has no cross-image reduction); value = N*256*K / t
where t is the MAX over ranks of the barrier-bracketed wall time of K steps.

The JSON line also carries
  roofline      the dominant kernel family (the implicit-GEMM contraction): algorithmic
                FLOPs of its launches in one forward / their summed HIP-event durations,
                measured on the library's stream in per-op instrumented forwards run
                right after the timed region (the timed region itself is uninstrumented);
                traffic / mfma_busy: HBM bytes per launch and the share of all SIMD cycles with
                a matrix pipe busy, from the committed rocprofv3 --pmc passes of this command,
                quoted only for the build they were measured on (source digest);
  hbm_kernels   the same for the bandwidth-bound kernels (GB/s against 8 TB/s): those of the timed
                (fused) forward plus, from a short leg in the one-kernel-per-reference-op mode run
                after the timed region, batch-norm / ReLU / add / max-pool;
  cpu_baseline  oracle/torch_port.py (a torch.nn.functional port of the reference's
                pytorch_inference.py, pinned by the golden logits) timed on this host's
                CPU cores on a bounded sample, rank 0 at N=1 only.
  dropin_route  (fp32, N=1 only; never `value`) the reference's own caller shape on this engine: the
                op-by-op graph of main.cu, one C-ABI call per reference op on NCHW tensors, on a deferred
                context (the C++ veneer's default) and literally -- the unchanged caller's rate next to
                the model driver's.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE = {"resnet50": 8.178368512, "resnet101": 15.60, "resnet152": 23.027253248}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak (spec)
PEAK_HBM_GBS = 8000.0         # HBM3E spec
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 matrix peak (spec, no sparsity)


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous batch split: rank r owns images [lo, hi)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_coord = {"rccl": False, "device": None, "mixed": False}


def init_dist(world: int, backend: str = "nccl"):
    """Process group for the barrier / max-over-ranks only; returns (rank, local_rank).
    The data path has no collective.  backend "nccl": RCCL for device tensors with gloo beside it for host
    tensors (one group, "cpu:gloo,cuda:nccl") -- agree_on_rccl() then decides which half carries the
    coordination; "gloo": host tensors only (ranks sharing one device in a rehearsal, CPU tests)."""
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        _coord["mixed"] = backend == "nccl"
        dist.init_process_group(backend="cpu:gloo,cuda:nccl" if backend == "nccl" else "gloo", rank=rank,
                                world_size=world)
    return rank, local_rank


def agree_on_rccl(world: int, device) -> str:
    """One all-reduce over RCCL on this rank's device.  If it works on EVERY rank (agreed through gloo) the
    barrier and the max over ranks run on device tensors over RCCL / xGMI, otherwise -- RCCL raising on a node
    where it cannot initialise -- on host tensors over gloo: the ranks still measure together, and the line
    says which it was (`world.backend`).  Returns "nccl" or "gloo"; None for one process."""
    if world == 1:
        return None
    import torch
    import torch.distributed as dist

    ok = 0.0
    if _coord["mixed"]:
        try:
            t = torch.ones(1, dtype=torch.float64, device=device)
            dist.all_reduce(t)
            torch.cuda.synchronize(device)
            ok = 1.0 if int(t.item()) == world else 0.0
        except Exception as e:  # noqa: BLE001 -- whatever RCCL raises, the host-side group still works
            print(f"bench: RCCL all-reduce failed on rank {dist.get_rank()} ({type(e).__name__}: {str(e)[:200]}); "
                  f"coordinating over gloo", file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # a host tensor: the gloo half
        ok = float(flag.item())
    _coord["rccl"], _coord["device"] = ok == 1.0, device
    return "nccl" if _coord["rccl"] else "gloo"


def _coord_tensor(value: float):
    """A 1-element tensor where the process group coordinates: this rank's device (RCCL) or the host (gloo)."""
    import torch

    return torch.tensor([value], dtype=torch.float64, device=_coord["device"] if _coord["rccl"] else None)


def barrier(world: int):
    if world > 1:
        import torch
        import torch.distributed as dist

        t = _coord_tensor(0.0)
        dist.all_reduce(t)
        if _coord["rccl"]:
            torch.cuda.synchronize(_coord["device"])


def max_over_ranks(value: float, world: int, device=None) -> float:
    if world == 1:
        return value
    import torch.distributed as dist

    t = _coord_tensor(value)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def device_locality(device_index: int, bind: bool):
    """Where this rank's device sits in the host (rn_device_locality: PCI address, NUMA node, local CPUs) and,
    with `bind`, this process moved onto those cores (the ones it may use; RN_SHARD_AFFINITY=0 turns it off):
    on a two-socket 8-GPU node half of the ranks would otherwise launch from the far socket."""
    import ctypes

    import resnet_c_amd as R

    pci, cpus, node = ctypes.create_string_buffer(32), ctypes.create_string_buffer(256), ctypes.c_int(-1)
    if R._lib.lib().rn_device_locality(device_index, pci, 32, ctypes.byref(node), cpus, 256) != 0:
        return {}
    out = {"pci": pci.value.decode(), "numa_node": node.value, "local_cpus": cpus.value.decode(), "bound": False}
    if bind and out["local_cpus"] and os.environ.get("RN_SHARD_AFFINITY", "1")[:1] != "0":
        local = set()
        for part in out["local_cpus"].split(","):
            a, _, b = part.partition("-")
            local |= set(range(int(a), int(b or a) + 1))
        both = local & os.sched_getaffinity(0)
        if both:
            try:
                os.sched_setaffinity(0, both)
                out["bound"] = True
            except OSError:
                pass
    out["cpus_allowed"] = len(os.sched_getaffinity(0))
    return out


def gather_ranks(me: dict, world: int):
    """Every rank's own record (rank, device, its own ms_per_step ...) on every rank, in rank order.
    Host data: pickled and gathered as host tensors (the gloo half of the group, whichever half coordinates)."""
    if world == 1:
        return [me]
    import pickle

    import torch
    import torch.distributed as dist

    n = dist.get_world_size()
    blob = pickle.dumps(me)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(n)]
    dist.all_gather(sizes, torch.tensor([len(blob)], dtype=torch.int64))
    cap = max(int(x.item()) for x in sizes)
    mine = torch.zeros(cap, dtype=torch.uint8)
    mine[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    bufs = [torch.zeros(cap, dtype=torch.uint8) for _ in range(n)]
    dist.all_gather(bufs, mine)
    return [pickle.loads(bytes(b[:int(x.item())].tolist())) for b, x in zip(bufs, sizes)]


def world_block(ranks, size, backend):
    """The bench line's `world` object: what torch.distributed saw, every rank's device and its OWN time
    for the K steps (before it waited for the others), so that a straggler -- a slow device, a host thread
    on the wrong socket -- is visible next to the max-over-ranks time the value is computed from."""
    out = {"size": size, "backend": backend, "ranks": ranks}
    times = [r["ms_per_step"] for r in ranks if "ms_per_step" in r]
    if times:
        out["slowest_rank_ms_per_step"] = max(times)
        out["fastest_rank_ms_per_step"] = min(times)
        out["slowest_rank"] = max(ranks, key=lambda r: r.get("ms_per_step", 0.0))["rank"]
    return out


def cpu_baseline(arch: str, state, seconds_budget: float = 30.0, engine=None, batch: int = 256):
    """Reference-equivalent PyTorch forward on the host CPU, bounded sample.  `engine`: a callable
    (NCHW fp32 array -> logits) of the GPU path; the port's logits on the sample's first images
    are held against it in this very run (the port itself is pinned to the reference module's
    golden logits by tests/test_oracle.py)."""
    import numpy as np
    import torch

    import resnet_c_amd as R
    from oracle import torch_port as TP

    t = TP.to_torch(state)
    # a 1-GPU box gives this process about 16 host cores; oversubscribing the 256-thread host
    # makes oneDNN slower, not faster (measured: 30 img/s at 16 threads, 10 img/s at 128)
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    # the bench batch itself (1 warm-up + 3 timed forwards of B = 256: about 35 s at
    # ~30 images/s); the deeper networks run a smaller batch to stay inside the same budget
    B = batch if arch == "resnet50" else max(32, batch // 4)
    x = torch.from_numpy(R.weights.generate_input(B, seed=123))
    with torch.no_grad():
        TP.resnet_forward(t, x[:2], arch)  # thread pool, oneDNN primitives
        t0 = time.perf_counter()
        y = TP.resnet_forward(t, x, arch)  # the warm-up forward at full size, also the time estimate
        one = time.perf_counter() - t0
        reps = int(max(1, min(3, seconds_budget / max(one, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(reps):
            y = TP.resnet_forward(t, x, arch)
        dt = time.perf_counter() - t0
    assert np.isfinite(y.numpy()).all()
    check = None
    if engine is not None:
        got = engine(x[:4].numpy())
        check = float(np.abs(got - y[:4].numpy()).max())
    return {"value": round(B * reps / dt, 2), "unit": "images/s", "cores": threads, "kind": "port",
            "max_abs_diff_vs_gpu_logits_on_4_images": check,
            "sample": f"1 warm-up + {reps} timed forwards of batch {B} ({arch} fp32, torch {torch.__version__} "
                      f"functional port of pytorch_inference.py, {threads} threads of "
                      f"{os.cpu_count()} host cpus)"}


def _committed_pmc(args, launches_per_forward: int, stem: str):
    """The newest profiles/round*/<stem><config tag>.json when it describes THIS build and THIS graph:
    (record, file, stale?).  Hardware counters cannot be read from inside this process; the committed
    rocprofv3 --pmc passes of this same command carry the digest of the kernel sources they were measured
    on and their launch count, and when either differs from this build / this run the figure no longer
    describes what ran: (None, file, True).  Configurations without a file: (None, None, False)."""
    if not (args.batch == 256 and args.mode == "fused" and args.arch == "resnet50") and \
            not (args.arch == "resnet152" and args.batch == 128 and args.mode == "fused" and args.dtype == "f32"):
        return None, None, False
    import glob

    import resnet_c_amd as R

    tag = "" if args.dtype == "f32" else f"_{args.dtype}"
    if args.arch != "resnet50":
        tag += f"_{args.arch}_b{args.batch}"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", f"{stem}{tag}.json")))
    if not files:
        return None, None, False
    src = os.path.relpath(files[-1], ROOT)
    try:
        rec = json.load(open(files[-1]))
        # the launch count moves by a few with the tuner's tile picks (a cut tail adds a finishing
        # launch): more than that is another graph
        if rec.get("source_digest") != R._lib.source_digest() or \
                abs(int(rec.get("launches", -100)) - int(launches_per_forward)) > 4:
            return None, src, True
        return rec, src, False
    except Exception:
        return None, src, True


def pmc_traffic(args, launches_per_forward: int):
    """(HBM bytes per launch of the contraction kernels, file they come from, stale?): the committed
    FETCH_SIZE / WRITE_SIZE passes (profiles/round*/final_hbm_traffic_pmc*.json, tools/pmc_traffic.py:
    separate passes, FETCH_SIZE x2 on gfx950); traffic = null, traffic_stale = true for any other build."""
    rec, src, stale = _committed_pmc(args, launches_per_forward, "final_hbm_traffic_pmc")
    try:
        return (round(rec["traffic_bytes_per_launch"]) if rec else None), src, stale
    except Exception:
        return None, src, True


def pmc_mfma_busy(args, launches_per_forward: int):
    """Matrix-pipe utilisation of the contraction launches from the committed SQ counter pass
    (profiles/round*/final_pmc_mfma_utilisation*.json, tools/pmc_mfma.py): {mfma_busy, ...}, file, stale?"""
    rec, src, stale = _committed_pmc(args, launches_per_forward, "final_pmc_mfma_utilisation")
    try:
        if rec:
            rec = {k: rec[k] for k in ("mfma_busy", "mfma_busy_in_busy_cu", "cu_busy", "wait_inst_over_wave_cycles",
                                       "wait_any_over_wave_cycles", "lds_bank_conflict_over_idx_active")}
        return rec, src, stale
    except Exception:
        return None, src, True


def summarize_profile(recs, n_forwards: int):
    """Per-op-family totals per forward from the instrumented forwards."""
    fam = {}
    for r in recs:
        f = fam.setdefault(r["op"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
        f["ms"] += r["ms"]
        f["flops"] += r["flops"]
        f["bytes"] += r["bytes"]
        f["launches"] += 1
    for f in fam.values():
        for k in ("ms", "flops", "bytes"):
            f[k] /= n_forwards
        f["launches"] //= n_forwards
    return fam


def host_pipeline(model, x_host, B, fused, steps):
    """PCIe-inclusive rate (never `value`): every batch is uploaded from pinned host memory
    and its logits downloaded, through rn_pipeline_* (copy stream + two slots), next to the
    same thing done strictly in sequence as the reference's main() does it."""
    import resnet_c_amd as R
    pipe = R.Pipeline(model, B, fused=fused)
    for _ in range(2):  # fill both staging buffers once; the producer is not what is timed
        pipe.input_buffer()[...] = x_host
        pipe.submit()
    pipe.collect(); pipe.collect()
    t0 = time.perf_counter()
    for _ in range(steps):
        if pipe.in_flight() == 2:
            pipe.collect()
        pipe.submit()
    while pipe.in_flight():
        pipe.collect()
    overlapped = B * steps / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit()
        pipe.collect()
    sequential = B * steps / (time.perf_counter() - t0)
    pipe.close()
    return {"value": round(overlapped, 1), "unit": "images/s",
            "sequential": round(sequential, 1),
            "what": "upload (pinned, %.0f MB/batch) + forward + logits download per batch; "
                    "overlapped = two slots in flight" % (x_host.nbytes / 1e6)}


def dropin_route(arch: str, state, x_dev, B: int, steps: int):
    """The reference's own caller shape on this engine (never `value`): createResnet / resnetForward, one C-ABI
    call per reference op on NCHW fp32 tensors (main.cu:127-226), with the context deferred (what the C++ veneer
    runs by default: conv + in-place bn / add / relu as one launch in the caller's buffers) and literally (one
    launch per call).  Rates of the unchanged caller next to the model driver's `value`."""
    import resnet_c_amd as R

    ctx = R.get_ctx()
    m = R.createResnet(arch, state)
    out = {}
    try:
        for name, deferred in (("literal", False), ("deferred", True)):
            ctx.set_deferred(deferred)
            ctx.set_weight_cache(True)
            for _ in range(2):
                R.resnetForward(m, x_dev)
                ctx.flush()
            ctx.sync()
            s0 = ctx.deferred_stats()
            n = max(2, min(steps, 5))
            t0 = time.perf_counter()
            for _ in range(n):
                R.resnetForward(m, x_dev)
                ctx.flush()
            ctx.sync()
            dt = (time.perf_counter() - t0) / n
            out[name] = {"images_per_s": round(B / dt, 1), "ms_per_forward": round(dt * 1e3, 3)}
            if deferred:
                s1 = ctx.deferred_stats()
                out[name]["launches_per_forward"] = {k: (s1[k] - s0[k]) // n
                                                     for k in ("fused_launches", "literal_launches", "transposes")}
    finally:
        ctx.set_deferred(False)
        ctx.set_weight_cache(False)
    out["what"] = ("the reference-shaped op-by-op graph (174 calls per ResNet-50 forward) on NCHW tensors: deferred = "
                   "rn_ctx_set_deferred(1), literal = one launch per call; B = %d, inputs resident" % B)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="resnet50", choices=sorted(GFLOP_PER_IMAGE))
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--mode", default="fused", choices=["fused", "ops"],
                    help="fused epilogues (default) or one kernel per reference op")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="storage type of activations/weights (accumulation is always fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the PCIe-inclusive leg")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the leg that runs the reference's op-by-op caller shape (fp32, N = 1 only)")
    ap.add_argument("--no-tune", action="store_true", help="skip the per-layer tile tuning pass")
    ap.add_argument("--streams", type=int, default=0, choices=[0, 1, 2, 4],
                    help="parts of a batch that run on streams of their own (they fill each other's kernel "
                         "tails); 0 = the library's default (2 for fp32 and for bf16 storage); the roofline "
                         "block and ms_per_step_one_stream are always taken on one stream")
    ap.add_argument("--no-ops-leg", action="store_true",
                    help="skip the short one-kernel-per-reference-op leg behind the timed region (fp32 fused "
                         "runs take the batch-norm / ReLU / add / max-pool GB/s of hbm_kernels from it); "
                         "profiler passes that pick 'the last forwards' of the process use this")
    ap.add_argument("--front-parts", type=int, default=1,
                    help="stem + max-pool + first stage in this many slices per batch part (Infinity-Cache reuse)")
    ap.add_argument("--profile-forwards", type=int, default=3)
    args = ap.parse_args()

    import numpy as np
    import torch

    import resnet_c_amd as R

    world = args.gpus
    ndev = torch.cuda.device_count()
    if world > 1:
        assert int(os.environ.get("WORLD_SIZE", "1")) == world, "launch with torch.distributed.run"
    # rehearsal on a box with fewer GPUs than ranks: ranks share devices, which RCCL refuses -- host-side group
    # (RN_BENCH_FORCE_RCCL=1 tries RCCL all the same: the rehearsal of the fall-back to gloo)
    shared = world > ndev and os.environ.get("RN_BENCH_FORCE_RCCL", "0")[:1] != "1"
    rank, local_rank = init_dist(world, backend="gloo" if shared else "nccl")
    device_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(device_index)
    R.set_device(device_index)
    ctx = R.get_ctx()

    # who takes part: the launcher's world size as torch.distributed sees it, and every rank's device
    me = {"rank": rank, "local_rank": local_rank, "device_index": device_index,
          "device": torch.cuda.get_device_name(device_index), "pid": os.getpid()}
    dist_world, dist_backend = 1, None
    if world > 1:
        import torch.distributed as dist

        dist_world, dist_backend = dist.get_world_size(), agree_on_rccl(world, torch.device("cuda", device_index))
    me.update(device_locality(device_index, bind=world > 1))
    print(f"bench: rank {rank}/{dist_world} ({dist_backend or 'single process'}) on cuda:{device_index} "
          f"{me['device']}", file=sys.stderr, flush=True)

    B = args.batch
    state = R.weights.generate_state(args.arch, seed=0)
    model = R.NativeModel(args.arch, state=state, ctx=ctx, dtype=args.dtype)
    cfg_streams = args.streams     # what was asked for: 0 = the library default
    if cfg_streams:
        model.set_streams(cfg_streams)
    parts = model.parts(B)         # what a forward of B images actually runs as (by dtype, B and the setting)
    model.set_front_parts(args.front_parts)
    lo, hi = shard_bounds(world * B, rank, world)
    # this rank's shard of the global batch: image i depends on (seed, i) only
    per = 3 * 224 * 224
    x_host = np.empty((B, 3, 224, 224), dtype=np.float32)
    for j, i in enumerate(range(lo, hi)):
        u = R.weights.uniform01("input", per, 0, offset=i * per)
        x_host[j] = (-2.0 + 4.0 * u).astype(np.float32).reshape(3, 224, 224)
    x_dev = R.FloatTensor.from_numpy(x_host, R.Device.GPU)
    logits = R.FloatTensor((B, 1000), R.Device.GPU)
    fused = args.mode == "fused"

    if not args.no_tune:
        # per-layer tile choice for this batch size (speed only; results are bit-identical)
        model.tune(x_dev.data(), B, logits.data(), fused)
    for _ in range(args.warmup):
        model.forward_ptr(x_dev.data(), B, logits.data(), fused)
    ctx.sync()

    barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.forward_ptr(x_dev.data(), B, logits.data(), fused)
    ctx.sync()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0     # this rank's own K steps, before it waits for the others
    barrier(world)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world, device=torch.device("cuda", device_index))
    # every rank's own time next to the max: a straggler (a slow device, a host thread on the wrong
    # socket) shows in the line instead of only stretching the job's time
    me["ms_per_step"] = round(own / args.steps * 1e3, 4)
    me["images_per_s"] = round(B * args.steps / own, 1)
    ranks = gather_ranks(me, world)

    out = logits.numpy()
    assert np.isfinite(out).all(), "non-finite logits"

    # the same steps with the whole batch on ONE stream (what the roofline block below describes)
    ms_one_stream = elapsed / args.steps * 1e3
    if parts > 1:
        model.set_streams(1)
        for _ in range(2):
            model.forward_ptr(x_dev.data(), B, logits.data(), fused)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            model.forward_ptr(x_dev.data(), B, logits.data(), fused)
        ctx.sync()
        ms_one_stream = (time.perf_counter() - t1) / args.steps * 1e3
        model.set_streams(cfg_streams)  # 0: back to the library default, exactly as the timed region ran
        assert model.parts(B) == parts
        assert np.array_equal(logits.numpy(), out), "one stream and several must give the same bits"

    # per-kernel durations: HIP events on the library's stream around every op
    model.set_profiling(True)
    recs = []
    lib = R._lib.lib()
    launches0 = lib.rn_ctx_launch_count(ctx.handle)
    for _ in range(args.profile_forwards):
        model.forward_ptr(x_dev.data(), B, logits.data(), fused)
        recs += model.profile()
    kernel_launches = (lib.rn_ctx_launch_count(ctx.handle) - launches0) // max(1, args.profile_forwards)
    fam = summarize_profile(recs, max(1, args.profile_forwards))
    # north_star asks for the achieved HBM GB/s of batch-norm / ReLU / add / max-pool: the timed
    # forward is fused and has none of them as kernels, so a short leg in the one-kernel-per-
    # reference-op mode (same weights, same batch, same process) measures them
    ops_fam = {}
    if fused and args.dtype == "f32" and not args.no_ops_leg:
        model.forward_ptr(x_dev.data(), B, logits.data(), False)  # untimed: first touch of its buffers
        orecs = []
        for _ in range(max(1, args.profile_forwards)):
            model.forward_ptr(x_dev.data(), B, logits.data(), False)
            orecs += model.profile()
        ops_fam = summarize_profile(orecs, max(1, args.profile_forwards))
    model.set_profiling(False)

    if rank != 0:
        return
    images = world * B * args.steps
    value = images / elapsed
    gemm_names = [k for k in fam if k.startswith("conv2d") or k == "linear"]
    g_flops = sum(fam[k]["flops"] for k in gemm_names)
    g_ms = sum(fam[k]["ms"] for k in gemm_names)
    g_bytes = sum(fam[k]["bytes"] for k in gemm_names)
    g_ops = sum(fam[k]["launches"] for k in gemm_names)
    # kernel launches of the contraction family: one per op record plus the launches that add the
    # K-chunk pieces of cut tail tiles (their time is inside the op's event bracket)
    other_ops = sum(f["launches"] for k, f in fam.items() if k not in gemm_names)
    g_launch = int(kernel_launches) - other_ops
    achieved = g_flops / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
    hbm = {}
    for k, f in fam.items():
        if k in gemm_names or f["ms"] <= 0:
            continue
        gbs = f["bytes"] / (f["ms"] * 1e-3) / 1e9
        hbm[k] = {"launches": f["launches"], "ms_per_forward": round(f["ms"], 4),
                  "GBps": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4)}
    for k in ("batchnorm2d", "relu", "add", "maxpool2d"):
        f = ops_fam.get(k)
        if f and f["ms"] > 0 and k not in hbm:
            gbs = f["bytes"] / (f["ms"] * 1e-3) / 1e9
            hbm[k] = {"launches": f["launches"], "ms_per_forward": round(f["ms"], 4), "GBps": round(gbs, 1),
                      "frac": round(gbs / PEAK_HBM_GBS, 4),
                      "from": "ops-mode leg (one kernel per reference op) behind the timed region"}
    peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    prec = "fp32" if args.dtype == "f32" else "bf16"
    g_gbps = g_bytes / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
    # what binds the contraction family as a whole: the larger of its two roofline fractions
    # (fp32: the matrix pipe; bf16 storage: 16x the matrix rate on half the bytes -> HBM)
    frac_mfma, frac_hbm = achieved / peak, g_gbps / PEAK_HBM_GBS
    bound = "mfma" if frac_mfma >= frac_hbm else "hbm"
    traffic, traffic_source, traffic_stale = pmc_traffic(args, g_launch)
    busy, busy_source, busy_stale = pmc_mfma_busy(args, g_launch)
    result = {
        "metric": "images/sec ResNet-50 224x224 fp32 batch=256"
                  if args.arch == "resnet50" and B == 256 and args.dtype == "f32"
                  else f"images/sec {args.arch} 224x224 {prec} batch={B}",
        "value": round(value, 2),
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "ms_per_step_one_stream": round(ms_one_stream, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic (seeded uniform [-2,2) images, generated weights in the reference's "
                "weights_bin format; inputs resident in HBM)",
        "config": {"workload": f"{args.arch} {prec} forward, batch {B} per GPU, 224x224 "
                               + ("(BASELINE.json configs[2])" if args.dtype == "f32" and args.arch == "resnet50"
                                  else "(BASELINE.json configs[3], per-GPU shard)" if args.dtype == "bf16"
                                  else "(BASELINE.json configs[4])"),
                   "global_batch": world * B, "batch_per_gpu": B,
                   "mode": "fused conv+bn+relu(+add) epilogues" if fused else "one kernel per reference op",
                   "streams_per_gpu": parts, "streams_configured": cfg_streams or "library default",
                   "front_parts": args.front_parts,
                   "parallelism": f"batch split over {world} GPU(s), weights replicated, no collective"},
        "roofline": {"bound": bound,
                     "achieved": round(achieved, 2) if bound == "mfma" else round(g_gbps, 1),
                     "peak": peak if bound == "mfma" else PEAK_HBM_GBS,
                     "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                     "frac": round(max(frac_mfma, frac_hbm), 4),
                     "traffic": traffic, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
                     # share of the chip's SIMD-cycles with a matrix pipe busy during the contraction launches
                     # (SQ_VALU_MFMA_BUSY_CYCLES; committed --pmc pass of this command, this build only)
                     "mfma_busy": busy["mfma_busy"] if busy else None,
                     "mfma_busy_detail": busy, "mfma_busy_source": busy_source, "mfma_busy_stale": busy_stale,
                     "source_digest": R._lib.source_digest(),
                     "kernel": "conv_gemm_kernel" + (" / conv_wide_kernel / conv_strip_kernel / chain_kernel" if args.dtype == "bf16" else " / chain32_kernel") +
                               " (implicit-GEMM conv2d + fc on " +
                               ("v_mfma_f32_32x32x2_f32)" if args.dtype == "f32" else "v_mfma_f32_32x32x16_bf16)"),
                     "mfma_TFLOPs": round(achieved, 2), "mfma_frac": round(frac_mfma, 4),
                     "hbm_GBps": round(g_gbps, 1), "hbm_frac": round(frac_hbm, 4),
                     "ops_per_forward": g_ops, "launches_per_forward": g_launch,
                     "avg_launch_us": round(g_ms * 1e3 / max(g_launch, 1), 2),
                     "flops_per_forward": g_flops, "bytes_per_forward": g_bytes,
                     "ms_per_forward": round(g_ms, 4),
                     "measured_on": "HIP events around every op of %d instrumented forwards right after the "
                                    "timed region, whole batch on ONE stream (with streams_per_gpu > 1 the timed "
                                    "region overlaps the kernels of the batch parts, so their sum exceeds "
                                    "ms_per_step)" % args.profile_forwards},
        "world": world_block(ranks, dist_world, dist_backend),
        "whole_step_mfma_frac": round(value / world * GFLOP_PER_IMAGE[args.arch] / 1e3 / peak, 4),
        "hbm_kernels": hbm,
    }
    if world == 1 and not args.no_pipeline:
        result["host_pipeline"] = host_pipeline(model, x_host, B, fused, args.steps)
    if world == 1 and args.dtype == "f32" and not args.no_dropin:
        result["dropin_route"] = dropin_route(args.arch, state, x_dev, B, args.steps)
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.arch, state, batch=B,
                                              engine=lambda a: model.forward(a, fused=fused))
        diff = result["cpu_baseline"]["max_abs_diff_vs_gpu_logits_on_4_images"]
        assert diff is not None and diff <= (1e-4 if args.dtype == "f32" else 0.25), diff
    print(json.dumps(result))


if __name__ == "__main__":
    main()
