"""Pixel-coordinate grid of the lip crop (reference: src/face_simple/rendering.py:9-28)."""
from __future__ import annotations

import torch


def get_coords(width, height, device, add_noise_uv=False, raw_noise_std=0.0):
    """Normalised (u, v) for every pixel, row-major, endpoints inclusive -> [H*W, 2].

    Same signature as the reference.  The grid is evaluated with torch.linspace on the HOST and
    then moved, so the fp32 values are bit-identical to the reference's CPU path on any device
    (the positional encoding multiplies them by up to 512, so last-ulp differences matter).
    """
    x = torch.linspace(0.0, 1.0, int(width))
    y = torch.linspace(0.0, 1.0, int(height))
    u = x.unsqueeze(0).expand(int(height), int(width))
    v = y.unsqueeze(1).expand(int(height), int(width))
    coords = torch.stack([u, v], -1).reshape(-1, 2).contiguous().to(device)
    if add_noise_uv:
        coords = coords + torch.randn(coords.shape, device=coords.device) * raw_noise_std
    return coords


_GRID_CACHE = {}


def shared_coords(width, height, device):
    """The same grid as `get_coords(width, height, device)`, built once per (size, device) and SHARED: for callers inside the package
    that only read it (a training step rebuilt and re-uploaded it several times: a host-to-device copy from pageable memory is a
    synchronisation each time)."""
    key = (int(width), int(height), str(torch.device(device)))
    grid = _GRID_CACHE.get(key)
    if grid is None:
        grid = _GRID_CACHE[key] = get_coords(width, height, device)
    return grid
