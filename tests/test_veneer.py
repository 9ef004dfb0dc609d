"""The C++ veneer (include/rn/*.hpp) with the reference's class names compiles against
librn_hip.so (CPU) and computes the same numbers as the oracle (GPU)."""
import os
import re
import subprocess

import numpy as np
import pytest

import resnet_c_amd as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name="veneer_smoke"):
    exe = str(tmp_path / name)
    libdir = os.path.dirname(R._lib.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", f"-I{ROOT}/include",
                    f"{ROOT}/examples/{name}.cpp", f"-L{libdir}", "-lrn_hip",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                   check=True, capture_output=True, text=True)
    return exe


def test_veneer_compiles_with_a_plain_host_compiler(tmp_path):
    assert os.path.exists(_build(tmp_path))
    assert os.path.exists(_build(tmp_path, "resnet_veneer"))


REF_MAIN = "/root/reference/cuda/inference/main.cu"


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("debug", [False, True])
def test_reference_caller_compiles_and_links_against_the_veneer(tmp_path, debug):
    """INTEGRATION.md section 2, literally: the reference's own cuda/inference/main.cu (read where it
    lies, never copied) compiles with a plain host compiler once its three includes resolve to
    the veneer -- here through three one-line forwarding headers, which is the include edit of
    the diff -- and links against librn_hip.so with no extra flags.  -DDEBUG switches on
    safeCudaMalloc's allocation log (helpers.cuh:28-33)."""
    for name in ("nn", "ops", "tensor"):
        (tmp_path / f"{name}.cuh").write_text(f'#include "rn/{name}.hpp"\n')
    libdir = os.path.dirname(R._lib.LIB_PATH)
    exe = str(tmp_path / "main_ref")
    cmd = ["g++", "-x", "c++", "-std=c++20", f"-I{tmp_path}", f"-I{ROOT}/include", REF_MAIN, "-o", exe,
           f"-L{libdir}", "-lrn_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    if debug:
        cmd.insert(1, "-DDEBUG")
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(exe)
    # the names the reference's callers use (helpers.cuh:6-35) are all defined by the veneer
    hdr = open(os.path.join(ROOT, "include", "rn", "tensor.hpp")).read()
    for name in ("inline void gpuAssert(int code, const char *file, int line, bool abort = true)",
                 "inline void *safeCudaMalloc(uint64_t size)", "#define gpuErrchk", "#define CEIL",
                 "#include <cassert>", "#include <iomanip>", "#include <numeric>"):
        assert name in hdr, name


MAIN_REF = os.path.join(ROOT, "oracle", "_ref", "main_ref")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(MAIN_REF),
                    reason="oracle/_ref/main_ref is built from /root/reference by `make -C oracle` in the build container")
def test_reference_main_runs_unchanged_on_the_engine(tmp_path, state152, finch, golden_dir):
    """The reference's own program (cuda/inference/main.cu, compiled by oracle/Makefile against
    include/rn/*.hpp, not a line changed) run as its author runs it: cwd holds weights_bin/ in
    the save_weights.py format and test_bins/ILSVRC2012_val_00004749.bin; it builds ResNet-152
    from the reference-named classes, runs every op through the C-ABI and prints the reference's
    'max index is N' (main.cu:243-251).  N must be the top-1 of the reference PyTorch module's
    golden logits on the same generated weights."""
    os.mkdir(tmp_path / "weights_bin")
    os.mkdir(tmp_path / "test_bins")
    R.weights.save_weights_bin(state152, str(tmp_path / "weights_bin"))
    finch.astype(np.float32).tofile(tmp_path / "test_bins" / "ILSVRC2012_val_00004749.bin")
    r = subprocess.run([MAIN_REF], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = int(np.load(os.path.join(golden_dir, "resnet152_finch_logits.npy")).argmax(1)[0])
    assert r.stdout.splitlines()[-1] == f"max index is {want}", r.stdout


@pytest.mark.gpu
def test_veneer_runs_and_matches_oracle(tmp_path):
    from oracle import oracle as O

    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    m = re.search(r"out shape \(2, 32, 6, 6\) checksum ([-0-9.e+]+)", r.stdout)
    assert m, r.stdout
    B, C, H, W = 2, 32, 6, 6
    w = (0.01 * (np.arange(C * C * 9) % 7) - 0.02).astype(np.float32).reshape(C, C, 3, 3)
    x = ((np.arange(B * C * H * W) % 5) - 2.0).astype(np.float32).reshape(B, C, H, W)
    ones, zeros = np.ones(C, np.float32), np.zeros(C, np.float32)
    y = O.conv2d(x, w, 1, 1)
    y = O.relu_(O.batchnorm2d_(y, ones, 0.5 * ones, zeros, ones))
    y = O.add_(y, x)
    assert abs(float(m.group(1)) - float(y.astype(np.float64).sum())) < 1e-2


@pytest.mark.gpu
def test_whole_network_through_the_veneer(tmp_path, state50, finch, golden_dir):
    """examples/resnet_veneer.cpp: ResNet-50 built from the reference-named C++ classes, loading
    weights_bin/<key> files, one veneer call (= one C-ABI call, NCHW, synchronous) per reference
    op -- the literal route (RN_VENEER_LITERAL=1; the deferred default is tests/test_defer_gpu.py's).
    Same logits as the reference module's goldens, same 'max index is N' line."""
    exe = _build(tmp_path, "resnet_veneer")
    os.mkdir(tmp_path / "weights_bin")
    R.weights.save_weights_bin(state50, str(tmp_path / "weights_bin"))
    x = np.concatenate([finch, R.weights.generate_input(1, seed=7)[:1]]).astype(np.float32)
    x.tofile(tmp_path / "input.bin")
    r = subprocess.run([exe, "50", "input.bin", "logits.bin"], cwd=tmp_path, capture_output=True,
                       text=True, timeout=300, env={**os.environ, "RN_VENEER_LITERAL": "1"})
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "logits.bin", dtype=np.float32).reshape(2, 1000)
    want = np.load(os.path.join(golden_dir, "resnet50_finch_logits.npy"))
    assert np.abs(got[:1] - want).max() <= 1e-4
    lines = [l for l in r.stdout.splitlines() if l.startswith("max index is")]
    assert lines == [f"max index is {int(got[0].argmax())}", f"max index is {int(got[1].argmax())}"]
    assert int(got[0].argmax()) == 112
    m = R.NativeModel("resnet50", state=state50)
    try:
        assert np.abs(m.forward(x, fused=False) - got).max() <= 1e-5
    finally:
        m.close()
