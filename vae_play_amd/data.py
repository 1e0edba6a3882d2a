"""Synthetic circle dataset of kungyao/vae-play's ``train.py`` (host side, no GPU work).

  generate_circle_param / generate_circle_img / encode_circle_param / decode_circle_param / generate_batch_circle
      <- tools/utils.py:13-71 (restated: that module imports cv2 / skimage, which this image lacks)
  CDataset, CDataset.train_collate_fn  <- datasets/dataset.py:23-93 (the generated mode ``ifGen=True``; the file mode
      reads ``./datas/*.png`` with PIL exactly as the reference does when PIL is importable)

Images are float32 (C, n, n) in [0, 1] like ``TF.to_tensor`` of the uint8 array; targets are
(log(r / n), (x - n/2) / (n/2), (y - n/2) / (n/2)).
"""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch
from torch.utils.data import Dataset

CHANNEL_SIZE = 1   # datasets/dataset.py:20


def generate_circle_param(n: int, min: int) -> Dict[str, int]:
    half_n = n // 2
    radius = np.random.randint(low=min, high=half_n - min)
    center_x = radius + np.random.randint(low=0, high=n - 2 * radius)
    center_y = radius + np.random.randint(low=0, high=n - 2 * radius)
    return {"radius": radius, "x": center_x, "y": center_y}


def generate_circle_img(n: int, x: int, y: int, radius: int, channel_size: int = 3) -> np.ndarray:
    sample = np.linspace(0, n - 1, n)
    xv, yv = np.meshgrid(sample, sample)
    res = (xv - x) ** 2 + (yv - y) ** 2
    img = np.where(res <= radius ** 2, 255, 0).astype(np.uint8)
    if channel_size == 3:
        img = np.stack([img, img, img], axis=-1)
    return img


def encode_circle_param(n: int, radius: torch.Tensor, center_x: torch.Tensor, center_y: torch.Tensor):
    half = n // 2
    return {"radius": torch.log(radius / n), "x": (center_x - half) / half, "y": (center_y - half) / half}


def decode_circle_param(n: int, c_radius: torch.Tensor, c_center_x: torch.Tensor, c_center_y: torch.Tensor):
    half = n // 2
    return {"radius": torch.exp(c_radius) * n, "x": c_center_x * half + half, "y": c_center_y * half + half}


def _to_tensor(img: np.ndarray) -> torch.Tensor:
    """torchvision's TF.to_tensor for a uint8 HxW or HxWxC array: CHW float32 / 255."""
    if img.ndim == 2:
        img = img[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).float().div(255)


def generate_batch_circle(n: int, radius: torch.Tensor, center_x: torch.Tensor, center_y: torch.Tensor, channel_size: int = 3):
    return torch.stack([_to_tensor(generate_circle_img(n, x.item(), y.item(), r.item(), channel_size=channel_size))
                        for r, x, y in zip(radius, center_x, center_y)], dim=0)


class CDataset(Dataset):
    def __init__(self, n: int, min_radius: int = 10, data_size: int = 4096, ifGen: bool = False, ifWrite: bool = False,
                 data_dir: str = "./datas"):
        self.n, self.ifGen, self.ifWrite, self.data_dir = n, ifGen, ifWrite, data_dir
        self.params = []
        if ifGen:
            self.imgs = None
            for _ in range(data_size):
                self.params.append(generate_circle_param(n, min_radius))
            self.data_size = data_size
        else:
            self.imgs = []
            for f in sorted(os.listdir(data_dir)):
                self.imgs.append(os.path.join(data_dir, f))
                _, r, x, y = f.split("_")
                self.params.append({"radius": int(r), "x": int(x), "y": int(y.split(".")[0])})
            self.data_size = len(self.imgs)

    def __len__(self):
        return self.data_size

    def __getitem__(self, idx):
        param = self.params[idx]
        if self.ifGen:
            img = generate_circle_img(self.n, param["x"], param["y"], param["radius"], channel_size=CHANNEL_SIZE)
            if self.ifWrite:
                from .imageio import write_png
                os.makedirs(self.data_dir, exist_ok=True)
                write_png(os.path.join(self.data_dir, f"{idx}_{int(param['radius'])}_{int(param['x'])}_{int(param['y'])}.png"), img)
            return _to_tensor(img), param
        from PIL import Image
        img = Image.open(self.imgs[idx], "r").convert("L" if CHANNEL_SIZE == 1 else "RGB")
        if img.size[0] > self.n:
            img = img.resize((self.n, self.n))
        return _to_tensor(np.asarray(img)), param

    @staticmethod
    def train_collate_fn(batch):
        imgs, params = zip(*batch)
        imgs = torch.stack(imgs, dim=0)
        img_size = imgs.shape[-1]
        rs = torch.FloatTensor([p["radius"] for p in params])
        xs = torch.FloatTensor([p["x"] for p in params])
        ys = torch.FloatTensor([p["y"] for p in params])
        enc = encode_circle_param(img_size, rs, xs, ys)
        return imgs, torch.stack([enc["radius"], enc["x"], enc["y"]], dim=-1)
