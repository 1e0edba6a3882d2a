"""Alias: ``models.blocks`` -> ``vae_play_amd.blocks`` (same class names, constructor signatures and state_dict keys as the
reference module of this name; every op underneath is a HIP kernel behind the C ABI)."""
from vae_play_amd.blocks import *  # noqa: F401,F403
from vae_play_amd import blocks as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
