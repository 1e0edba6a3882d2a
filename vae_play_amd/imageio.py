"""PNG output without torchvision / cv2: an image-grid writer with the call surface of
``torchvision.utils.save_image(tensor, path, nrow=, padding=, pad_value=)`` as train.py:100-106 uses it, on zlib."""
from __future__ import annotations

import struct
import zlib

import numpy as np
import torch


def write_png(path: str, img: np.ndarray) -> None:
    """uint8 array (H, W) grey or (H, W, 3) RGB -> PNG file (8-bit, no interlace, filter 0)."""
    if img.dtype != np.uint8 or img.ndim not in (2, 3):
        raise ValueError("write_png expects a uint8 array of shape (H, W) or (H, W, 3)")
    if img.ndim == 3 and img.shape[2] == 1:
        img = img[:, :, 0]
    h, w = img.shape[:2]
    color = 0 if img.ndim == 2 else 2
    raw = b"".join(b"\x00" + np.ascontiguousarray(img[r]).tobytes() for r in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def make_grid(tensor: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """(B, C, H, W) -> (3 or C, rows*(H+padding)+padding, cols*(W+padding)+padding), torchvision's layout: nrow images
    per row, single-channel images replicated to 3 channels."""
    if tensor.dim() == 3:
        tensor = tensor.unsqueeze(0)
    tensor = tensor.detach().float().cpu()
    if tensor.size(1) == 1:
        tensor = tensor.repeat(1, 3, 1, 1)
    n = tensor.size(0)
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    H, W = tensor.size(2) + padding, tensor.size(3) + padding
    grid = tensor.new_full((tensor.size(1), H * ymaps + padding, W * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * H + padding:(y + 1) * H, x * W + padding:(x + 1) * W] = tensor[k]
            k += 1
    return grid


def save_image(tensor: torch.Tensor, path: str, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> None:
    grid = make_grid(tensor, nrow=nrow, padding=padding, pad_value=pad_value)
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    write_png(path, arr)
