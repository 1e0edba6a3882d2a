"""``state_dict`` checkpoints.  The reference pickles whole modules and optimisers
(``torch.save({"networks": networks, "optims": optims, "epoch": epoch})``, train.py:154-161), which ties a checkpoint
to the defining source files and executes code on load.  Here a checkpoint is a flat dict of tensors and numbers:
loadable with ``torch.load(..., weights_only=True)``, and interchangeable with the reference's classes because the
drop-in modules use the same ``state_dict`` keys."""
from __future__ import annotations

from typing import Dict, Optional

import torch


def save_checkpoint(path: str, networks: Dict[str, torch.nn.Module], optims: Optional[Dict[str, object]] = None, epoch: int = 0):
    ckpt = {"format": "vae_play_amd/1", "epoch": int(epoch),
            "networks": {k: {n: t.detach().cpu() for n, t in m.state_dict().items()} for k, m in networks.items()},
            "optims": {k: o.state_dict() for k, o in (optims or {}).items()}}
    torch.save(ckpt, path)


def load_checkpoint(path: str, networks: Dict[str, torch.nn.Module], optims: Optional[Dict[str, object]] = None) -> int:
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if ckpt.get("format") != "vae_play_amd/1":
        raise ValueError(f"{path}: not a vae_play_amd checkpoint")
    for k, m in networks.items():
        m.load_state_dict(ckpt["networks"][k], strict=True)
    for k, o in (optims or {}).items():
        if k in ckpt["optims"]:
            o.load_state_dict(ckpt["optims"][k])
    return int(ckpt["epoch"])
