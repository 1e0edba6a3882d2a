"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# BASELINE.json north_star: outputs within 1e-3 relative (fp32) of the reference CPU path.
NORTH_STAR_RTOL = 1e-3
# The fp32-MFMA path is much tighter than that; single ops are held to this so that an indexing
# bug cannot hide inside the 1e-3 budget.
OP_RTOL = 2e-5


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| relative to the tensor's scale (max|b|)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


ERRORS = {}   # what -> worst relative error seen this session (dumped by conftest at session end)


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    key = f"{os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0]}::{what}"
    ERRORS[key] = max(ERRORS.get(key, 0.0), e)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def record(what: str, err: float) -> float:
    """Log a relative error for the session report without asserting."""
    key = f"{os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0]}::{what}"
    ERRORS[key] = max(ERRORS.get(key, 0.0), float(err))
    return err
