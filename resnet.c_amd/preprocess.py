"""JPEG -> ``test_bins/<stem>.bin`` (raw fp32 [1,3,224,224] NCHW).

Build-owned equivalent of the reference's ``convert_imgs_to_bin.py:12-23``,
which applies torchvision's ``ResNet152_Weights.IMAGENET1K_V1.transforms()``
and dumps the tensor with ``struct.pack('f', ...)``.  torchvision is not
available offline, so the preset is restated with PIL + numpy:

    resize so the short side is 256 (bilinear, PIL's antialiased reducer),
    long side = int(256 * long / short); centre-crop 224x224 with
    round((size - 224) / 2) offsets; uint8 -> fp32 / 255;
    (x - mean) / std with mean (0.485, 0.456, 0.406), std (0.229, 0.224, 0.225).

The real-weights top-1 of the reference's test image is not recorded anywhere
in the reference, so this restatement is "parity unpinned" against torchvision
itself; what is pinned is every downstream result on the tensor it produces.
"""
from __future__ import annotations

import os

import numpy as np

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def preprocess_image(path: str, resize: int = 256, crop: int = 224) -> np.ndarray:
    from PIL import Image

    with Image.open(path) as im:
        im = im.convert("RGB")
        w, h = im.size
        if w <= h:
            nw, nh = resize, int(resize * h / w)
        else:
            nw, nh = int(resize * w / h), resize
        im = im.resize((nw, nh), Image.BILINEAR)
        left = int(round((nw - crop) / 2.0))
        top = int(round((nh - crop) / 2.0))
        im = im.crop((left, top, left + crop, top + crop))
        px = np.asarray(im, dtype=np.uint8)
    x = px.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(MEAN, dtype=np.float32)) / np.asarray(STD, dtype=np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1)[None], dtype=np.float32)


def convert_dir(input_dir: str, out_dir: str) -> list:
    """Every ``*.jpeg`` in input_dir -> ``out_dir/<stem>.bin``; returns the paths."""
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for name in sorted(os.listdir(input_dir)):
        if name.endswith(".jpeg") and os.path.isfile(os.path.join(input_dir, name)):
            dst = os.path.join(out_dir, os.path.splitext(name)[0] + ".bin")
            preprocess_image(os.path.join(input_dir, name)).tofile(dst)
            written.append(dst)
    return written


def load_bin(path: str, batch: int = 1, hw: int = 224) -> np.ndarray:
    """Read a test_bins file back as [batch,3,hw,hw] (main.cu:236-237)."""
    x = np.fromfile(path, dtype=np.float32)
    return x.reshape(batch, 3, hw, hw)
