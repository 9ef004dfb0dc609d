"""N > 1 path on CPU: two gloo ranks run bench.py's sharding / barrier / max-over-ranks
plumbing.  The forward itself has no collective; what must be
right is that the ranks own disjoint contiguous shards that cover the global batch,
that each rank's images are the same ones a single process would generate, and that
the reported time is the maximum over ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    import resnet_c_amd as R
    from oracle import oracle as O

    r, lr = bench.init_dist(world, backend="gloo")
    assert (r, lr) == (rank, rank)
    per_gpu, hw = 3, 16
    lo, hi = bench.shard_bounds(world * per_gpu, rank, world)
    x = np.stack([R.weights.generate_input(1, seed=0, hw=hw)[0] * 0 + i for i in range(lo, hi)])
    # a stand-in "forward" with no cross-image reduction, run by the CPU checker
    w = R.weights.generate_tensor("conv1.weight", (4, 3, 3, 3), 0)
    y = O.conv2d(x.astype(np.float32), w, 1, 1).reshape(hi - lo, -1).sum(1)
    # a host-side group (this test, rehearsals with ranks sharing a device): coordination over gloo, and it says so
    assert bench.agree_on_rccl(world, None) == "gloo" and bench.agree_on_rccl(1, None) is None
    bench.barrier(world)
    t = bench.max_over_ranks(1.0 + rank, world)
    # the per-rank records of the bench line's `world` block: every rank's own time next to the max
    me = {"rank": rank, "local_rank": rank, "device_index": rank, "device": f"cpu{rank}", "pid": os.getpid(),
          "ms_per_step": 10.0 + 5.0 * rank, "images_per_s": 256e3 / (10.0 + 5.0 * rank)}
    wb = bench.world_block(bench.gather_ranks(me, world), dist.get_world_size(), dist.get_backend())
    assert wb["size"] == world and wb["backend"] == "gloo" and [r["rank"] for r in wb["ranks"]] == [0, 1]
    assert [r["ms_per_step"] for r in wb["ranks"]] == [10.0, 15.0] and wb["ranks"][rank]["pid"] == os.getpid()
    assert (wb["slowest_rank"], wb["slowest_rank_ms_per_step"], wb["fastest_rank_ms_per_step"]) == (1, 15.0, 10.0)
    gathered = [torch.zeros(per_gpu, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(y.astype(np.float64)))
    q.put((rank, lo, hi, t, torch.cat(gathered).numpy()))
    dist.destroy_process_group()


def test_two_rank_shards_cover_the_batch_and_time_is_max():
    world, port = 2, 29000 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, all0), (r1, lo1, hi1, t1, all1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 6)
    assert t0 == t1 == 2.0  # max over ranks of (1.0, 2.0)
    assert np.array_equal(all0, all1)
    # same rows a single process computes for the whole batch
    import resnet_c_amd as R
    from oracle import oracle as O
    x = np.stack([R.weights.generate_input(1, seed=0, hw=16)[0] * 0 + i for i in range(6)])
    w = R.weights.generate_tensor("conv1.weight", (4, 3, 3, 3), 0)
    want = O.conv2d(x.astype(np.float32), w, 1, 1).reshape(6, -1).sum(1)
    np.testing.assert_allclose(all0, want, rtol=0, atol=0)


def test_shard_bounds_properties():
    import bench

    for total in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 4, 8):
            spans = [bench.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert bench.shard_bounds(2048, 3, 8) == (768, 1024)
