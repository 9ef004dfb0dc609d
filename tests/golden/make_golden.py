#!/usr/bin/env python3
"""Generate the golden vectors in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (it reads /root/reference, which does not
exist on the GPU box):

    python tests/golden/make_golden.py

What it does
------------
The reference's PyTorch script (pytorch_inference.py) cannot be imported as a
module: it imports torchvision (not installed) and, at module level, loads a
file, moves to "cuda" and downloads pretrained weights.  Its three model
definitions -- ``ResnetBlock``, ``make_layer``, ``Resnet152`` -- are pure
torch.nn, so this script parses the file with ``ast``, compiles ONLY those three
definitions (no other statement of the file is executed) and instantiates them
on the CPU.  ResNet-50 / ResNet-101 are the same class with ``layer1..4`` rebuilt by the
reference's own ``make_layer`` with 3/4/6/3 and 3/4/23/3 blocks.

Weights come from the build's deterministic generator
(resnet_c_amd.weights.generate_state, seed 0) -- pretrained weights cannot be
fetched offline -- and are loaded with ``load_state_dict``; the state_dict keys
of the reference module are the weights_bin file names.

Outputs (small, committed):
  finch_224.bin                 preprocessed test image, raw fp32 [1,3,224,224]
  <arch>_finch_logits.npy       fp32 logits [1,1000] of the reference module
  <arch>_finch_logits_f64.npy   the same module run in float64 (adjudicator)
  <arch>_rand2_logits.npy       logits for generate_input(2, seed=7)
  <arch>_taps.json              per-stage mean / mean-abs (float64) + top-1s
  ops_kat.npz                   per-op known answers on the reference's own
                                test.cu input patterns (torch.nn.functional)
"""
import ast
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import resnet_c_amd  # noqa: E402
from resnet_c_amd import preprocess, weights  # noqa: E402


def load_reference_classes():
    src = open(os.path.join(REF, "pytorch_inference.py")).read()
    tree = ast.parse(src)
    wanted = {"ResnetBlock", "make_layer", "Resnet152"}
    body = [n for n in tree.body
            if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in wanted]
    assert {n.name for n in body} == wanted
    mod = ast.Module(body=body, type_ignores=[])
    ns = {"torch": torch, "nn": nn, "F": F}
    exec(compile(mod, "pytorch_inference.py[model classes]", "exec"), ns)
    return ns


def build(ns, arch):
    m = ns["Resnet152"](1000)
    if arch != "resnet152":
        d = weights.depths_of(arch)
        mk = ns["make_layer"]
        m.layer1 = mk(64, 64, 256, n_blocks=d[0])
        m.layer2 = mk(256, 128, 512, n_blocks=d[1], stride=2)
        m.layer3 = mk(512, 256, 1024, n_blocks=d[2], stride=2)
        m.layer4 = mk(1024, 512, 2048, n_blocks=d[3], stride=2)
    state = weights.generate_state(arch, seed=0)
    sd = {k: torch.from_numpy(v.copy()) for k, v in state.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("num_batches_tracked") for k in missing), missing
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == weights.param_count(arch), (n_params, weights.param_count(arch))
    m.eval()
    return m


def taps_of(m, x):
    rec = {}

    def hook(name):
        def f(_mod, _inp, out):
            o = out.detach().double()
            rec[name] = {"mean": float(o.mean()), "mean_abs": float(o.abs().mean())}
        return f

    hs = [getattr(m, n).register_forward_hook(hook(n))
          for n in ("maxpool", "layer1", "layer2", "layer3", "layer4", "avgpool")]
    with torch.no_grad():
        y = m(x)
    for h in hs:
        h.remove()
    return y, rec


def ops_kat():
    """Known answers on the input patterns of the reference's cuda/test.cu."""
    out = {}
    # conv2dTest (test.cu:6-17,34-36): B=2, Cin=1, Cout=2, k=2, 7x7, arange data
    x = torch.arange(2 * 1 * 7 * 7, dtype=torch.float32).view(2, 1, 7, 7)
    w = torch.arange(2 * 1 * 2 * 2, dtype=torch.float32).view(2, 1, 2, 2)
    out["conv_x"], out["conv_w"] = x.numpy(), w.numpy()
    out["conv_y"] = F.conv2d(x, w).numpy()
    # linearTest (test.cu:100-136): B=3, 16->8, arange data
    x = torch.arange(3 * 16, dtype=torch.float32).view(3, 16)
    w = torch.arange(8 * 16, dtype=torch.float32).view(8, 16)
    b = torch.arange(8, dtype=torch.float32)
    out["lin_x"], out["lin_w"], out["lin_b"] = x.numpy(), w.numpy(), b.numpy()
    out["lin_y"] = F.linear(x, w, b).numpy()
    # reluTest (test.cu:178-185): N=17, alternating sign
    x = torch.tensor([(i if i % 2 == 0 else -i) for i in range(17)], dtype=torch.float32)
    out["relu_x"], out["relu_y"] = x.numpy(), F.relu(x).numpy()
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ns = load_reference_classes()
    finch = preprocess.preprocess_image(
        os.path.join(REF, "test_imgs", "ILSVRC2012_val_00004749.jpeg"))
    finch.tofile(os.path.join(HERE, "finch_224.bin"))
    rand2 = weights.generate_input(2, seed=7)
    for arch in ("resnet50", "resnet101", "resnet152"):
        m = build(ns, arch)
        y, rec = taps_of(m, torch.from_numpy(finch))
        np.save(os.path.join(HERE, f"{arch}_finch_logits.npy"), y.numpy())
        with torch.no_grad():
            y2 = m(torch.from_numpy(rand2))
        np.save(os.path.join(HERE, f"{arch}_rand2_logits.npy"), y2.numpy())
        m64 = m.double()
        with torch.no_grad():
            y64 = m64(torch.from_numpy(finch).double())
            y64r = m64(torch.from_numpy(rand2).double())
        np.save(os.path.join(HERE, f"{arch}_finch_logits_f64.npy"), y64.numpy())
        np.save(os.path.join(HERE, f"{arch}_rand2_logits_f64.npy"), y64r.numpy())
        srt = np.sort(y64.numpy()[0])[::-1]
        info = {
            "arch": arch, "seed": 0, "torch": torch.__version__,
            "finch_top1": int(y.argmax(1)[0]), "finch_top1_f64": int(y64.argmax(1)[0]),
            "finch_top2_gap_f64": float(srt[0] - srt[1]),
            "rand2_top1": [int(v) for v in y2.argmax(1)],
            "max_abs_f32_vs_f64": float(np.abs(y.numpy() - y64.numpy()).max()),
            "logit_abs_max": float(np.abs(y64.numpy()).max()),
            "taps": rec,
        }
        with open(os.path.join(HERE, f"{arch}_taps.json"), "w") as f:
            json.dump(info, f, indent=1, sort_keys=True)
        print(arch, {k: v for k, v in info.items() if k != "taps"})
        print("   taps", rec)
    np.savez(os.path.join(HERE, "ops_kat.npz"), **ops_kat())


if __name__ == "__main__":
    main()
