"""Colour jitter of the camera image (2D augmentation of the reference's loaders).

The reference builds ``torchvision.transforms.ColorJitter(*color_jitter)`` (nuscenes_dataloader.py:197, semantic_kitti.py:267,
a2d2.py:242, virtual_kitti_dataloader.py:164; the shipped configs pass ``[0.4, 0.4, 0.4]`` = brightness, contrast,
saturation) and applies it to the PIL image after the point rasterisation and before the float conversion
(nuscenes_dataloader.py:286-287).  torchvision is not in this image, so this is a restatement of its PIL code path:

  * one ``torch.randperm(4)`` for the order of the four operations, then one ``torch.empty(1).uniform_(lo, hi)`` per ACTIVE
    operation in the fixed order brightness, contrast, saturation, hue (torch RNG, not numpy's: the loaders' numpy draws
    are not disturbed);
  * brightness / contrast / saturation = ``PIL.ImageEnhance.{Brightness, Contrast, Color}(img).enhance(factor)``;
    hue = shift of the H channel of the HSV image by ``uint8(factor * 255)`` with wrap-around.

Parity note: the arithmetic of this class is unpinned (no torchvision here to generate vectors from); its place in the
pipeline and its arguments are pinned by tests/golden/loader_*.npz.
"""
from __future__ import annotations

import numbers

import numpy as np
import torch


def _interval(value, name, center=1.0, bound=(0.0, float("inf")), clip_first_on_zero=True):
    """A number v means [center - v, center + v] (lower end clipped at 0 for the multiplicative factors); a pair is taken
    as given; the degenerate interval [center, center] switches the operation off (None)."""
    if isinstance(value, numbers.Number):
        if value < 0:
            raise ValueError(f"If {name} is a single number, it must be non negative.")
        value = [center - float(value), center + float(value)]
        if clip_first_on_zero:
            value[0] = max(value[0], 0.0)
    elif isinstance(value, (tuple, list)) and len(value) == 2:
        value = [float(value[0]), float(value[1])]
        if not bound[0] <= value[0] <= value[1] <= bound[1]:
            raise ValueError(f"{name} values should be between {bound}")
    else:
        raise TypeError(f"{name} should be a single number or a list/tuple with length 2.")
    return None if value[0] == value[1] == center else value


class ColorJitter:
    def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
        self.brightness = _interval(brightness, "brightness")
        self.contrast = _interval(contrast, "contrast")
        self.saturation = _interval(saturation, "saturation")
        self.hue = _interval(hue, "hue", center=0.0, bound=(-0.5, 0.5), clip_first_on_zero=False)

    def draw(self):
        """(order of the four operations, brightness, contrast, saturation, hue factors or None), torch RNG."""
        order = torch.randperm(4).tolist()
        f = [None if iv is None else float(torch.empty(1).uniform_(iv[0], iv[1])) for iv in
             (self.brightness, self.contrast, self.saturation, self.hue)]
        return order, f

    def __call__(self, img):
        from PIL import Image, ImageEnhance

        order, f = self.draw()
        for op in order:
            if f[op] is None:
                continue
            if op == 0:
                img = ImageEnhance.Brightness(img).enhance(f[0])
            elif op == 1:
                img = ImageEnhance.Contrast(img).enhance(f[1])
            elif op == 2:
                img = ImageEnhance.Color(img).enhance(f[2])
            else:
                mode = img.mode
                if mode in ("L", "1", "I", "F"):
                    continue
                h, s, v = img.convert("HSV").split()
                hh = np.array(h, dtype=np.uint8)
                with np.errstate(over="ignore"):
                    hh += np.uint8(int(f[3] * 255) % 256)
                img = Image.merge("HSV", (Image.fromarray(hh, "L"), s, v)).convert(mode)
        return img
