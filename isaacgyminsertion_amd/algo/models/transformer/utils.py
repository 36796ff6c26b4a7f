"""Tactile image transforms with the reference's names (algo/models/transformer/utils.py:131-278,
457-466): ``define_tactile_transforms`` -> (train transform, eval transform), ``TactileTransform``,
``GaussianNoise``, ``Masking``, ``set_seed``.

The reference builds these from torchvision (absent here) and applies them one image at a time in a Python
loop (utils.py:145-149).  Here each transform is a batched torch op over ``(n, C, W, H)`` on whatever
device the batch lives on:
  * deterministic part -- resize to (width, height) [identity at the configured 32x64], centre / random
    crop, patch masking, additive Gaussian noise: same arithmetic as the reference;
  * random photometric / geometric augmentation of the TRAIN transform (brightness+contrast jitter 0.1 with
    p 0.3, 5x5 Gaussian blur sigma in (0.01, 0.1) with p 0.5, rotation <= 3 degrees with p 0.5) --
    same distributions, drawn per image from torch's generator; not sample-identical to torchvision's RNG
    stream (no parity claim is made on augmented samples).
"""
import math
import os
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class GaussianNoise(nn.Module):
    """utils.py:457-466"""

    def __init__(self, std=0.1):
        super().__init__()
        self.std = std

    def forward(self, x):
        return x + torch.randn_like(x) * self.std if self.training else x


class Masking(nn.Module):
    """utils.py:191-216: zero whole patches with probability ``img_masking_prob`` (same mask for every
    channel of an image)."""

    def __init__(self, img_patch_size, img_masking_prob):
        super().__init__()
        self.img_patch_size, self.img_masking_prob = img_patch_size, img_masking_prob

    def forward(self, x):
        p = self.img_patch_size
        gh, gw = x.shape[-2] // p, x.shape[-1] // p
        drop = torch.rand((x.shape[0], gh, gw), device=x.device) < self.img_masking_prob
        keep = (~drop).to(x.dtype).repeat_interleave(p, 1).repeat_interleave(p, 2)
        out = x.clone()
        out[:, :, :gh * p, :gw * p] *= keep.unsqueeze(1)
        return out


class _Resize(nn.Module):
    def __init__(self, size):
        super().__init__()
        self.size = tuple(size)

    def forward(self, x):
        if tuple(x.shape[-2:]) == self.size:
            return x
        return F.interpolate(x, size=self.size, mode='bilinear', align_corners=False, antialias=True)


class _Crop(nn.Module):
    def __init__(self, size, random_offset):
        super().__init__()
        self.size, self.random_offset = tuple(size), random_offset

    def forward(self, x):
        h, w = x.shape[-2:]
        ch, cw = self.size
        if (h, w) == (ch, cw):
            return x
        if self.random_offset and self.training:
            top = int(torch.randint(0, h - ch + 1, (1,)))
            left = int(torch.randint(0, w - cw + 1, (1,)))
        else:
            top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
        return x[..., top:top + ch, left:left + cw]


class _TrainAugment(nn.Module):
    """brightness/contrast jitter, 5x5 Gaussian blur, small rotation; each drawn per image."""

    def forward(self, x):
        if not self.training:
            return x
        n, dev = x.shape[0], x.device
        # ColorJitter(brightness=0.1, contrast=0.1), p = 0.3
        on = (torch.rand(n, 1, 1, 1, device=dev) < 0.3).to(x.dtype)
        b = 1 + (torch.rand(n, 1, 1, 1, device=dev) * 0.2 - 0.1) * on
        c = 1 + (torch.rand(n, 1, 1, 1, device=dev) * 0.2 - 0.1) * on
        x = (x * b).clamp(0, 1) * on + x * (1 - on)
        mean = x.mean(dim=(1, 2, 3), keepdim=True)
        x = ((x - mean) * c + mean).clamp(0, 1) * on + x * (1 - on)
        # GaussianBlur(5, sigma in (0.01, 0.1)), p = 0.5
        on = (torch.rand(n, device=dev) < 0.5)
        sigma = torch.rand(n, device=dev) * 0.09 + 0.01
        t = torch.arange(-2, 3, device=dev, dtype=x.dtype)
        k1 = torch.exp(-0.5 * (t[None, :] / sigma[:, None]) ** 2)
        k1 = k1 / k1.sum(1, keepdim=True)
        ident = torch.zeros(5, device=dev, dtype=x.dtype)
        ident[2] = 1
        k1 = torch.where(on[:, None], k1, ident[None, :])
        C = x.shape[1]
        xp = F.pad(x, (2, 2, 2, 2), mode='reflect').reshape(1, n * C, x.shape[2] + 4, x.shape[3] + 4)
        kh = k1.repeat_interleave(C, 0)
        xp = F.conv2d(xp, kh[:, None, :, None], groups=n * C)
        xp = F.conv2d(xp, kh[:, None, None, :], groups=n * C)
        x = xp.reshape(n, C, *x.shape[2:])
        # RandomRotation(3 degrees), p = 0.5 (nearest resampling, zero fill: torchvision's defaults)
        on = (torch.rand(n, device=dev) < 0.5).to(x.dtype)
        ang = (torch.rand(n, device=dev) * 6 - 3) * on * (math.pi / 180)
        cos, sin = torch.cos(ang), torch.sin(ang)
        H, W = x.shape[-2:]
        theta = torch.stack([torch.stack([cos, -sin * H / W, torch.zeros_like(cos)], 1),
                             torch.stack([sin * W / H, cos, torch.zeros_like(cos)], 1)], 1)
        grid = F.affine_grid(theta, list(x.shape), align_corners=False)
        return F.grid_sample(x, grid, mode='nearest', padding_mode='zeros', align_corners=False)


def define_tactile_transforms(width, height, crop_width, crop_height, img_patch_size=16, img_gaussian_noise=0.0,
                              img_masking_prob=0.0):
    """utils.py:219-277 -> (transform, eval_transform), both ``nn.Module``s over (n, C, W, H)."""
    stages = [_Resize((width, height)), _Crop((crop_width, crop_height), True), _TrainAugment()]
    if img_gaussian_noise > 0.0:
        stages.append(GaussianNoise(img_gaussian_noise))
    if img_masking_prob > 0.0:
        stages.append(Masking(img_patch_size, img_masking_prob))
    transform = nn.Sequential(*stages)
    eval_transform = nn.Sequential(_Resize((width, height)), _Crop((crop_width, crop_height), False))
    eval_transform.eval()
    return transform, eval_transform


class TactileTransform:
    """utils.py:131-156: (B, T, fingers, C, H, W) -> transformed, same rank.  One batched call instead
    of the reference's per-image loop."""

    def __init__(self, tactile_transform=None):
        self.tactile_transform = tactile_transform

    def __call__(self, tac_input):
        B, T, Fg, C, H, W = tac_input.shape
        y = tac_input.reshape(-1, C, H, W)
        if self.tactile_transform is not None:
            y = self.tactile_transform(y)
        return y.reshape(B, T, Fg, C, *y.shape[2:])


def set_seed(seed, torch_deterministic=False, rank=0):
    """utils.py:159-189"""
    if seed == -1 and torch_deterministic:
        seed = 42 + rank
    elif seed == -1:
        seed = np.random.randint(0, 10000)
    else:
        seed = seed + rank
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def log_output(*args, **kwargs):
    """utils.py:648-678 renders diagnostic figures of a batch; plotting is outside the scope table
     -- kept as a callable no-op so ``Runner.train`` keeps its call sites."""
    return None
