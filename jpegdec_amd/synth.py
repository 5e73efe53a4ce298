"""Deterministic synthetic baseline-JPEG inputs (SURVEY.md 8d recipe).

Multi-octave value noise (cell size from max(W,H)/4 halving down to 4 px, amplitude x0.7 per
octave, bicubic upsample), normalised to 0..255, plus N(0, 3^2) pixel noise, encoded with
Pillow/libjpeg-turbo as baseline Huffman JPEG (quality 85, Annex-K tables unless optimize=True).
There is no network on the build or GPU boxes, so every test/bench input comes from here (or
from the small fixtures committed under tests/golden/).
"""
import io

import numpy as np


def _upsample(grid: np.ndarray, width: int, height: int, fast: bool) -> np.ndarray:
    if fast:
        import torch
        import torch.nn.functional as F

        t = torch.from_numpy(grid)[None, None]
        return F.interpolate(t, size=(height, width), mode="bicubic", align_corners=False)[0, 0].numpy()
    from PIL import Image

    return np.asarray(Image.fromarray(grid, mode="F").resize((width, height), Image.BICUBIC),
                      dtype=np.float32)


def value_noise_image(width: int, height: int, channels: int = 3, seed: int = 1234,
                      fast: bool = None) -> np.ndarray:
    """fast=True upsamples with torch (multi-threaded) instead of Pillow; default: only for
    images above 4 Mpixel (the two bicubic kernels differ slightly, so `fast` is part of the
    recipe: small test images always use Pillow)."""
    if fast is None:
        fast = width * height > (1 << 22)
    rng = np.random.default_rng(seed)
    acc = np.zeros((channels, height, width), dtype=np.float32)
    cell = max(width, height) // 4
    amp = 1.0
    while cell >= 4:
        gw = max(2, (width + cell - 1) // cell + 1)
        gh = max(2, (height + cell - 1) // cell + 1)
        for c in range(channels):
            grid = rng.random((gh, gw), dtype=np.float32)
            acc[c] += amp * _upsample(grid, width, height, fast)
        amp *= 0.7
        cell //= 2
    lo = acc.min()
    hi = acc.max()
    acc = (acc - lo) / max(hi - lo, 1e-6) * 255.0
    acc += rng.normal(0.0, 3.0, size=acc.shape).astype(np.float32)
    out = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    if channels == 1:
        return out[0]
    return np.ascontiguousarray(np.transpose(out, (1, 2, 0)))


def encode_jpeg(pixels: np.ndarray, quality: int = 85, subsampling="4:2:0", optimize: bool = False,
                restart_rows: int = 0, restart_blocks: int = 0) -> bytes:
    """pixels: HxW (gray) or HxWx3 (RGB) uint8 -> baseline JPEG bytes."""
    from PIL import Image

    im = Image.fromarray(pixels)
    kw = dict(format="JPEG", quality=quality, optimize=optimize, progressive=False)
    if pixels.ndim == 3:
        kw["subsampling"] = {"4:4:4": 0, "4:2:2": 1, "4:2:0": 2}[subsampling]
    if restart_rows:
        kw["restart_marker_rows"] = restart_rows
    if restart_blocks:
        kw["restart_marker_blocks"] = restart_blocks
    buf = io.BytesIO()
    im.save(buf, **kw)
    return buf.getvalue()


def synth_jpeg(width: int, height: int, subsampling="4:2:0", seed: int = 1234, quality: int = 85,
               optimize: bool = False, restart_rows: int = 0, restart_blocks: int = 0) -> bytes:
    """subsampling: '4:2:0' | '4:4:4' | '4:2:2' | 'gray'."""
    ch = 1 if subsampling == "gray" else 3
    px = value_noise_image(width, height, ch, seed)
    return encode_jpeg(px, quality, subsampling if ch == 3 else None, optimize, restart_rows,
                       restart_blocks)


def bits_per_pixel(jpeg: bytes, width: int, height: int) -> float:
    return 8.0 * len(jpeg) / float(width * height)
