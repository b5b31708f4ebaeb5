"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases(dtype="f64"):
    """Fixtures of one precision ("f64": the reference as shipped; "f32": its fp32 build)."""
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as fh:
        return [c for c in json.load(fh)["cases"] if c.get("dtype", "f64") == dtype]


def load_golden(case):
    z = np.load(os.path.join(GOLDEN_DIR, case["name"] + ".npz"))
    return np.asfortranarray(z["f"]), np.asfortranarray(z["flux"])


def to_dev(a, device="cuda:0"):
    """Fortran-ordered numpy array -> torch tensor with reversed axes (same bytes)."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(a.T)).to(device)


def to_host(t):
    """Inverse of to_dev."""
    return np.asfortranarray(t.cpu().numpy().T)


def run_hip(M, inp, device="cuda:0"):
    """Device-resident HIP call on a dict of numpy inputs; returns (f, flux) numpy."""
    import torch
    d = {k: to_dev(v, device) for k, v in inp.items()}
    M.advect_scalar2D(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"])
    torch.cuda.synchronize()
    return to_host(d["f"]), to_host(d["flux"])


def max_abs(a, b):
    return float(np.max(np.abs(a - b)))
