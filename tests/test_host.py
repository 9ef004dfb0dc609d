"""Host-side logic that needs no GPU: layer table, generator, file formats, C-ABI surface."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import resnet_c_amd as R
from resnet_c_amd import _lib as L
from resnet_c_amd import weights as Wt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layer_table_matches_reference_counts():
    # param counts confirmed on the reference's nn.Module
    assert Wt.param_count("resnet50") == 25_557_032
    assert Wt.param_count("resnet152") == 60_192_808
    assert len(Wt.conv_specs("resnet50")) == 53
    assert len(Wt.conv_specs("resnet152")) == 155
    keys = [k for k, _ in Wt.tensor_specs("resnet152")]
    assert len(keys) == len(set(keys)) == 155 * 5 + 2
    assert "layer3.35.bn3.running_var" in keys and "layer1.0.downsample.1.bias" in keys
    assert Wt.bn_of("layer2.0.downsample.0") == "layer2.0.downsample.1"
    assert Wt.bn_of("layer4.2.conv3") == "layer4.2.bn3" and Wt.bn_of("conv1") == "bn1"
    flops = 0
    hw = 224
    for name, cin, cout, k, s, p in Wt.conv_specs("resnet50"):
        pass  # spatial sizes checked through the engine's own profile in the gpu tests
    with pytest.raises(ValueError):
        Wt.depths_of("resnet18")


def test_generator_is_deterministic_and_counter_based():
    a = Wt.generate_tensor("layer1.0.conv1.weight", (64, 64, 1, 1), seed=0)
    b = Wt.generate_tensor("layer1.0.conv1.weight", (64, 64, 1, 1), seed=0)
    c = Wt.generate_tensor("layer1.0.conv1.weight", (64, 64, 1, 1), seed=1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.dtype == np.float32 and np.abs(a).max() <= np.sqrt(6 / 64) + 1e-7
    # known values pin the hash itself (any change would silently invalidate the goldens)
    w = Wt.generate_tensor("conv1.weight", (64, 3, 7, 7), seed=0)
    np.testing.assert_allclose(w[0, 0, 0, :3], [0.01821303, 0.09906029, -0.19276875], rtol=0, atol=1e-8)
    # image i depends on (seed, i) only: batch 3 == three batch-1 calls at offsets
    x3 = Wt.generate_input(3, seed=5, hw=8)
    x1 = Wt.generate_input(1, seed=5, hw=8)
    assert np.array_equal(x3[:1], x1) and not np.array_equal(x3[0], x3[1])
    v = Wt.generate_tensor("bn1.running_var", (64,), 0)
    assert v.min() >= 0.5 and v.max() < 1.5


def test_weights_bin_round_trip(tmp_path):
    state = {k: Wt.generate_tensor(k, s, 3) for k, s in Wt.tensor_specs("resnet50")[:12]}
    d = tmp_path / "weights_bin"
    Wt.save_weights_bin(state, str(d))
    # an export also contains files the loader never reads (save_weights.py dumps every key)
    np.zeros(1, dtype=np.float32).tofile(d / "bn1.num_batches_tracked")
    for k, v in state.items():
        raw = np.fromfile(d / k, dtype=np.float32)
        assert raw.size == v.size and np.array_equal(raw, v.reshape(-1))
        assert os.path.getsize(d / k) == 4 * v.size  # headerless
    with pytest.raises((ValueError, FileNotFoundError, OSError)):
        Wt.load_weights_bin("resnet50", str(d))  # most keys missing


def test_preprocessed_fixture_shape_and_range(finch):
    assert finch.shape == (1, 3, 224, 224) and finch.dtype == np.float32
    # ImageNet normalisation range: (0-0.485)/0.229 .. (1-0.406)/0.225
    assert finch.min() >= -2.1180 and finch.max() <= 2.6401
    ref_img = "/root/reference/test_imgs/ILSVRC2012_val_00004749.jpeg"
    if os.path.exists(ref_img):  # build container only
        again = R.preprocess.preprocess_image(ref_img)
        assert np.array_equal(again, finch)


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rn_hip.h")).read()
    return sorted(set(re.findall(r"RN_API[^;(]*?\b(rn_[a-z0-9_]+)\s*\(", text)))


def test_c_abi_exports_every_declared_symbol():
    names = _declared_symbols()
    assert len(names) >= 45
    lib = L.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rn_hip.h but not exported"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature"
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True,
                         check=True).stdout
    exported = set(re.findall(r" T (rn_[a-z0-9_]+)", out))
    assert set(names) <= exported
    # nothing torch-typed or C++-mangled leaks out of the boundary
    assert not [s for s in re.findall(r" T (\S+)", out) if s.startswith("_Z")]


def test_c_abi_pure_host_entry_points():
    lib = L.lib()
    assert lib.rn_conv_output_size(224, 7, 2, 3) == 112
    assert lib.rn_conv_output_size(56, 3, 2, 1) == 28
    assert lib.rn_conv2d_input_channels(3) == 4 and lib.rn_conv2d_input_channels(64) == 64
    assert lib.rn_conv2d_packed_weight_numel(3, 64, 7) == 64 * 7 * 32
    assert lib.rn_conv2d_packed_weight_numel(64, 256, 3) == 256 * 9 * 64
    assert lib.rn_status_string(0) == b"ok"
    assert b"gfx950" in lib.rn_version()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "resnet.c_amd")
    bad = re.compile(r"(^\s*(import|from)\s+oracle\b)|liboracle|rn_oracle_", re.M)
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), f"{f}: the product path must not use the oracle"


def test_shape_and_cpu_tensor_mirror_reference_semantics(tmp_path):
    s = R.Shape((2, 3, 4))
    assert s.numel() == 24 and repr(s) == "(2, 3, 4)" and s.as_tuple(3) == (2, 3, 4)
    with pytest.raises(ValueError):
        s.as_tuple(4)
    t = R.Tensor(R.Device.GPU)  # empty tensor: shape (0,), falsy (tensor.cuh:62-65,222-225)
    assert not t and t.shape() == R.Shape((0,)) and t.data() is None
    a = R.Tensor.from_numpy(np.arange(6, dtype=np.float32))
    v = a.view((2, 3))
    assert v.shape() == R.Shape((2, 3)) and v.data() == a.data()  # view shares storage
    with pytest.raises(AssertionError):
        a.view((4, 2))
    p = tmp_path / "t.bin"
    a.save(str(p))
    b = R.Tensor.loadToCpu(str(p))
    assert b.shape() == R.Shape((6,)) and np.array_equal(b.numpy(), np.arange(6, dtype=np.float32))
    with pytest.raises(FileNotFoundError):
        R.Tensor.loadToCpu(str(tmp_path / "missing.bin"))
    c = R.Tensor((0,), R.Device.CPU)
    c.move_from(b)
    assert c and not b and b.shape() == R.Shape((0,))


def test_shard_bounds_of_the_c_driver_match_bench():
    """rn_shard_bounds (the plain-C multi-device driver) and bench.shard_bounds (the
    torch.distributed launch) split a batch the same way: contiguous, covering, in rank order."""
    import ctypes

    import bench
    lib = L.lib()
    for B in (1, 5, 7, 255, 256, 2048, 2049):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
                lib.rn_shard_bounds(B, r, world, ctypes.byref(lo), ctypes.byref(hi))
                assert (lo.value, hi.value) == bench.shard_bounds(B, r, world)
                assert lo.value == prev and hi.value - lo.value in (B // world, B // world + 1)
                prev = hi.value
            assert prev == B


def test_bench_quotes_pmc_traffic_only_for_the_build_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from committed rocprofv3 --pmc passes; the file carries the digest of the
    kernel sources and its launch count, and bench.py must say null + stale for any other build."""
    import argparse
    import json

    import bench

    d = tmp_path / "profiles" / "round9"
    d.mkdir(parents=True)
    digest = L.source_digest()
    assert re.fullmatch(r"[0-9a-f]{16}", digest) and digest == L.source_digest()
    rec = {"traffic_bytes_per_launch": 123456789.4, "launches": 69, "source_digest": digest}
    (d / "final_hbm_traffic_pmc.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    args = argparse.Namespace(arch="resnet50", batch=256, mode="fused", dtype="f32")
    assert bench.pmc_traffic(args, 69) == (123456789, os.path.join("profiles", "round9", "final_hbm_traffic_pmc.json"), False)
    assert bench.pmc_traffic(args, 71)[0] == 123456789          # the tuner's tile picks move the count by a few
    assert bench.pmc_traffic(args, 90)[0] is None and bench.pmc_traffic(args, 90)[2] is True
    rec["source_digest"] = "0" * 16                              # measured on other kernel sources
    (d / "final_hbm_traffic_pmc.json").write_text(json.dumps(rec))
    value, src, stale = bench.pmc_traffic(args, 69)
    assert value is None and stale is True and src.endswith("final_hbm_traffic_pmc.json")
    # the matrix-pipe utilisation pass is quoted under the same rule
    busy = {"mfma_busy": 0.71, "mfma_busy_in_busy_cu": 0.83, "cu_busy": 0.9, "wait_inst_over_wave_cycles": 0.6,
            "wait_any_over_wave_cycles": 0.2, "lds_bank_conflict_over_idx_active": 0.0,
            "launches": 69, "source_digest": digest}
    (d / "final_pmc_mfma_utilisation.json").write_text(json.dumps(busy))
    got, src, stale = bench.pmc_mfma_busy(args, 70)
    assert got["mfma_busy"] == 0.71 and src.endswith("final_pmc_mfma_utilisation.json") and stale is False
    busy["source_digest"] = "1" * 16
    (d / "final_pmc_mfma_utilisation.json").write_text(json.dumps(busy))
    assert bench.pmc_mfma_busy(args, 70)[0] is None and bench.pmc_mfma_busy(args, 70)[2] is True
    args.dtype = "bf16"                                          # no file for this configuration
    assert bench.pmc_traffic(args, 45) == (None, None, False)
    assert bench.pmc_mfma_busy(args, 45) == (None, None, False)
    args = argparse.Namespace(arch="resnet101", batch=64, mode="fused", dtype="f32")
    assert bench.pmc_traffic(args, 10) == (None, None, False)


def test_committed_pmc_traffic_is_that_of_the_committed_kernel_sources():
    """The traffic files under the newest profiles/roundN must describe HEAD's kernels: a kernel change
    without a fresh PMC pass would make the driver-run bench line say traffic_stale."""
    import glob
    import json

    for stem in ("final_hbm_traffic_pmc", "final_hbm_traffic_pmc_bf16", "final_pmc_mfma_utilisation",
                 "final_pmc_mfma_utilisation_bf16"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", stem + ".json")))
        assert files, stem
        rec = json.load(open(files[-1]))
        assert rec["source_digest"] == L.source_digest(), f"{files[-1]}: re-run tools/evidence.sh pmc and commit profiles/"


def test_bench_world_block_and_locality_degrade_gracefully():
    """bench.py's per-rank records: without a device the locality query says nothing (no exception, no
    binding); the world block carries every rank's own time and names the slowest."""
    import bench

    loc = bench.device_locality(0, bind=False)
    assert loc == {} or {"pci", "numa_node", "local_cpus", "bound", "cpus_allowed"} <= set(loc)
    ranks = [{"rank": 0, "ms_per_step": 17.0}, {"rank": 1, "ms_per_step": 19.5}, {"rank": 2, "ms_per_step": 16.9}]
    wb = bench.world_block(ranks, 3, "nccl")
    assert (wb["slowest_rank"], wb["slowest_rank_ms_per_step"], wb["fastest_rank_ms_per_step"]) == (1, 19.5, 16.9)
    assert bench.world_block([{"rank": 0}], 1, None) == {"size": 1, "backend": None, "ranks": [{"rank": 0}]}
    assert bench.gather_ranks({"rank": 0}, 1) == [{"rank": 0}]

