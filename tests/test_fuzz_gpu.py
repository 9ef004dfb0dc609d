"""Short runs of the randomised parity programs (tests/fuzz/conv_fuzz.py, ops_fuzz.py, model_fuzz.py) with
fixed seeds: every convolution entry point over random shapes / layouts / storage types / tile
candidates against the CPU oracle, the element-wise / pooling / batch-norm / linear entry points, and
the model driver's state machine (no switch that only reschedules the arithmetic may change a bit).
The long runs (minutes, other seeds) are started by hand; these keep the paths exercised in every run
of the suite.  Each tool exits non-zero with the failing case in its assertion message."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_tool(name, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", name), *args], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"{name} {' '.join(args)}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return r.stdout


def test_conv_fuzz_short():
    out = run_tool("conv_fuzz.py", "--seconds", "8", "--seed", "101")
    assert "all within tolerance" in out


def test_ops_fuzz_short():
    out = run_tool("ops_fuzz.py", "--seconds", "5", "--seed", "102")
    assert "bit-exact" in out


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_model_fuzz_short(dtype):
    out = run_tool("model_fuzz.py", "--seconds", "8", "--seed", "103", "--dtype", dtype)
    assert "gave the pool's bits" in out


def test_defer_fuzz_short():
    """Random programs of the seven reference ops on a deferred context (folding and non-folding chains, reads,
    partial reads, rn_observe, writes into operands, frees, parameter updates) against the literal run."""
    out = run_tool("defer_fuzz.py", "--seconds", "10", "--seed", "104")
    assert "all within tolerance" in out

