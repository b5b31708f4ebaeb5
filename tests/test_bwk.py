"""Second mini-app: biharmonic_wk_scalar of atmosphere/biharmonic_wk_kernel.F90.
CPU part: the oracle against the reference's own outputs (golden fixtures) and the C-ABI of
libbwk_hip.so.  GPU part: the HIP kernel through the C-ABI against the oracle -- EXACT variant
bit-identical, FAST variant to 1e-13 in the reference's own L2 metric (:69-73)."""
import ctypes
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

from util import GOLDEN_DIR, to_dev, to_host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "codesign-kernels_amd", "libbwk_hip.so")


@pytest.fixture(scope="module")
def B():
    from oracle import bwk
    bwk.build_lib()
    return bwk


def bwk_cases():
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as fh:
        return json.load(fh)["bwk"]["cases"]


@pytest.mark.parametrize("case", bwk_cases(), ids=lambda c: c["name"])
def test_oracle_matches_reference_golden_bitwise(B, case):
    inp = B.make_inputs(case["nelemd"], case["nlev"], case["qsize"])   # the reference's own LCG inputs
    h = hashlib.sha256(b"".join(inp[k].tobytes(order="F") for k in ("dvv", "elem", "qtens"))).hexdigest()
    assert h == case["inputs_sha256"]
    out = B.biharmonic(inp)
    assert hashlib.sha256(out.tobytes(order="F")).hexdigest() == case["qtens_out_sha256"]
    if case["stored"]:
        ref = np.asfortranarray(np.load(os.path.join(GOLDEN_DIR, case["name"] + ".npz"))["qtens"])
        assert np.array_equal(out, ref)


def test_reference_binary_still_agrees(B):
    if B.ref_exe(2) is None:
        pytest.skip("no oracle/_ref/bwk_ref_ne2")
    inp, out_ref, _ = B.run_reference(2)
    mine = B.make_inputs(2)
    assert all(np.array_equal(inp[k], mine[k]) for k in inp)
    assert np.array_equal(B.biharmonic(mine), out_ref)
    assert np.array_equal(B.biharmonic(mine, nthreads=2), out_ref)


def test_c_abi_exports_and_argument_errors():
    hdr = open(os.path.join(ROOT, "include", "bwk_hip.h")).read()
    names = set(re.findall(r"\b(bwk_[a-z0-9_]+)\s*\(", hdr))
    assert {"bwk_biharmonic_wk_scalar", "bwk_biharmonic_wk_scalar_device", "bwk_set_variant",
            "bwk_algorithmic_bytes", "bwk_last_error"} <= names
    if not os.path.exists(LIB):
        pytest.skip("libbwk_hip.so not built")
    L = ctypes.CDLL(LIB)
    for n in names:
        assert hasattr(L, n), n
    syms = subprocess.run(["nm", "-D", LIB], capture_output=True, text=True).stdout
    assert "oracle" not in syms                          # the product never links the oracle
    L.bwk_biharmonic_wk_scalar_device.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    L.bwk_last_error.restype = ctypes.c_char_p
    assert L.bwk_biharmonic_wk_scalar_device(0, 72, 40, None, None, None, None) == -1
    assert L.bwk_biharmonic_wk_scalar_device(4, 72, 40, None, None, None, None) == -1
    assert b"null" in L.bwk_last_error()
    assert L.bwk_biharmonic_wk_scalar_device(70000, 72, 40, None, None, None, None) == -2
    L.bwk_algorithmic_bytes.restype = ctypes.c_int64
    L.bwk_algorithmic_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    assert L.bwk_algorithmic_bytes(16, 72, 40) == 2 * 8 * 16 * 72 * 40 * 16 + 8 * (144 * 16 + 16)


# ------------------------------------------------------------------------------- GPU
def run_hip(K, inp):
    import torch
    d = {k: to_dev(v) for k, v in inp.items()}
    K.biharmonic_wk_scalar(d["elem"], d["qtens"], d["dvv"])
    torch.cuda.synchronize()
    return to_host(d["qtens"])


@pytest.fixture(scope="module")
def K():
    import torch
    assert torch.cuda.is_available()
    import codesign_kernels_amd.bwk as bwk_hip
    yield bwk_hip
    bwk_hip.set_variant(bwk_hip.VARIANT_EXACT)


@pytest.mark.gpu
@pytest.mark.parametrize("case", bwk_cases(), ids=lambda c: c["name"])
def test_hip_exact_reproduces_reference_golden_bitwise(K, B, case):
    K.set_variant(K.VARIANT_EXACT)
    out = run_hip(K, B.make_inputs(case["nelemd"], case["nlev"], case["qsize"]))
    assert hashlib.sha256(out.tobytes(order="F")).hexdigest() == case["qtens_out_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 72, 40), (5, 7, 3), (2, 16, 4), (7, 1, 129), (4, 26, 11), (33, 5, 5)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_hip_matches_oracle_on_other_sizes_and_data(K, B, shape, variant):
    """Sizes the reference never runs (slab counts that are not multiples of a workgroup pass,
    a single slab), signed random data."""
    K.set_variant(variant)
    inp = B.random_inputs(*shape, seed=sum(shape))
    ref = B.biharmonic(inp, nthreads=4)
    out = run_hip(K, inp)
    if variant == 0:
        assert np.array_equal(out, ref), f"max|d|={np.abs(out - ref).max():.3e}"
    else:
        assert B.l2norm(out, ref) < 1e-13


@pytest.mark.gpu
def test_hip_host_call(K, B):
    K.set_variant(K.VARIANT_EXACT)
    inp = B.make_inputs(3)
    ref = B.biharmonic(inp)
    q = inp["qtens"].copy(order="F")
    K.biharmonic_wk_scalar_host(inp["elem"], q, inp["dvv"])
    assert np.array_equal(q, ref)
    with pytest.raises(K.BwkError):
        K.biharmonic_wk_scalar_host(inp["elem"][:, :2].copy(order="F"), q, inp["dvv"])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_hip_large_sampled_against_oracle(K, B, variant):
    """5400 elements (a cubed-sphere ne=30 mesh) of random data on the device; sampled
    elements against the oracle (elements are independent, reference :193-199)."""
    import torch
    K.set_variant(variant)
    nelemd, nlev, qsize = 5400, 72, 40
    g = torch.Generator(device="cuda:0").manual_seed(7)
    q = torch.rand((nelemd, qsize, nlev, 4, 4), dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    el = torch.rand((nelemd, 144), dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    dv = torch.rand((4, 4), dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q0 = {ie: q[ie].clone() for ie in (0, 1234, 5399)}
    K.biharmonic_wk_scalar(el, q, dv)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(q).all())
    for ie, qin in q0.items():
        inp = {"dvv": to_host(dv), "elem": to_host(el[ie:ie + 1]), "qtens": to_host(qin[None])}
        ref = B.biharmonic(inp)
        out = to_host(q[ie:ie + 1])
        if variant == 0:
            assert np.array_equal(out, ref)
        else:
            assert B.l2norm(out, ref) < 1e-13
