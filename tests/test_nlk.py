"""Third mini-app ("after that"): the MPAS-Ocean high-order flux loop nest of
nested_loops/nested.F90.  CPU part: the oracle against the reference program's refFlx (golden
fixture; the shipped namelist size against the reference binary when it is present) and the
C-ABI of libnlk_hip.so.  GPU part: the HIP kernel through the C-ABI against the oracle -- EXACT
bit-identical; FAST within the program's own tolerance errTol = 1e-10 relative
(nested_vars.F90:36), in fact ~1e-16."""
import ctypes
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

from util import GOLDEN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "codesign-kernels_amd", "libnlk_hip.so")


@pytest.fixture(scope="module")
def N():
    from oracle import nlk
    nlk.build_lib()
    return nlk


def nlk_cases():
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as fh:
        return json.load(fh)["nlk"]["cases"]


def load_small():
    z = np.load(os.path.join(GOLDEN_DIR, "nlk_ref_small.npz"))
    inp = {k: (np.asfortranarray(z[k]) if z[k].ndim else z[k].item()) for k in z.files if k != "refFlx"}
    inp["nVertLevels"] = int(inp["nVertLevels"])
    inp["coef3rdOrder"] = float(inp["coef3rdOrder"])
    return inp, np.asfortranarray(z["refFlx"])


def test_oracle_matches_reference_golden_bitwise(N):
    inp, ref = load_small()
    assert inp["coef3rdOrder"] == N.coef3rd() == float(np.float32(2.14))   # fp32 literal, nested_vars.F90:35
    case = [c for c in nlk_cases() if c["name"] == "nlk_ref_small"][0]
    assert hashlib.sha256(ref.tobytes(order="F")).hexdigest() == case["refFlx_sha256"]
    assert np.array_equal(N.high_order_flux(inp), ref)


def test_reference_binary_still_agrees_at_the_shipped_size(N):
    if not os.path.exists(N.REF_EXE):
        pytest.skip("no oracle/_ref/nlk_ref")
    case = [c for c in nlk_cases() if c["name"] == "nlk_ref_nml"][0]
    inp, ref = N.run_reference(case["nEdges"], case["nCells"], case["nVertLevels"], case["nAdv"], seed=case["seed"])
    assert hashlib.sha256(ref.tobytes(order="F")).hexdigest() == case["refFlx_sha256"]
    assert np.array_equal(N.high_order_flux(inp), ref)


def test_c_abi_exports_and_argument_errors():
    hdr = open(os.path.join(ROOT, "include", "nlk_hip.h")).read()
    names = set(re.findall(r"\b(nlk_[a-z0-9_]+)\s*\(", hdr))
    assert {"nlk_high_order_flux", "nlk_high_order_flux_device", "nlk_set_variant", "nlk_algorithmic_bytes"} <= names
    if not os.path.exists(LIB):
        pytest.skip("libnlk_hip.so not built")
    L = ctypes.CDLL(LIB)
    for n in names:
        assert hasattr(L, n), n
    assert "oracle" not in subprocess.run(["nm", "-D", LIB], capture_output=True, text=True).stdout
    L.nlk_high_order_flux_device.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 9 + [ctypes.c_double] + [ctypes.c_void_p] * 2
    L.nlk_last_error.restype = ctypes.c_char_p
    args = [None] * 9 + [2.14, None, None]
    assert L.nlk_high_order_flux_device(0, 5, 5, 5, 2, *args) == -1
    assert L.nlk_high_order_flux_device(5, 5, 8, 7, 2, *args) == -1      # nvldim < nVertLevels
    assert L.nlk_high_order_flux_device(5, 5, 5, 5, 2, *args) == -1 and b"null" in L.nlk_last_error()


# ------------------------------------------------------------------------------- GPU
def to_dev(inp, K):
    import torch
    d = {}
    for k in K.INT_KEYS:
        d[k] = torch.from_numpy(np.ascontiguousarray(np.asarray(inp[k], dtype=np.int32).T)).to("cuda:0")
    for k in K.REAL_KEYS:
        d[k] = torch.from_numpy(np.ascontiguousarray(np.asarray(inp[k], dtype=np.float64).T)).to("cuda:0")
    return d


def run_hip(K, inp, fill=-3.5):
    import torch
    d = to_dev(inp, K)
    nvldim, nEdges = inp["normalThicknessFlux"].shape
    out = torch.full((nEdges, nvldim), fill, dtype=torch.float64, device="cuda:0")
    K.high_order_flux(d, inp["nVertLevels"], inp["coef3rdOrder"], out)
    torch.cuda.synchronize()
    return np.asfortranarray(out.cpu().numpy().T)


@pytest.fixture(scope="module")
def K():
    import torch
    assert torch.cuda.is_available()
    import codesign_kernels_amd.nlk as nlk_hip
    yield nlk_hip
    nlk_hip.set_variant(nlk_hip.VARIANT_EXACT)
    nlk_hip.set_kernel(-1)


@pytest.mark.gpu
def test_hip_exact_reproduces_reference_golden_bitwise(K):
    K.set_variant(K.VARIANT_EXACT)
    inp, ref = load_small()
    assert np.array_equal(run_hip(K, inp), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(25600, 2800, 100, 10, None), (1, 1, 1, 1, None), (37, 11, 64, 3, 64), (130, 40, 65, 7, 72),
                                   (500, 70, 100, 10, 104), (9, 300, 200, 12, 200), (64, 5, 3, 2, 8)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
@pytest.mark.parametrize("kernel", [0, 1, 2], ids=["one-level-per-lane", "two-levels-per-lane", "two-levels-vector-bookkeeping"])
def test_hip_matches_oracle(K, N, shape, variant, kernel):
    """The shipped namelist size; level counts around the wave width; padded leading dimension
    (nvldim > nVertLevels: padding rows stay untouched); ragged nAdvCellsForEdge, minLevelCell > 1,
    masked levels (the kernel takes them from the range check of a per-cell buffer descriptor and adds
    exact zeros instead of skipping), negative zero fluxes.  Both kernel forms (nlk_set_kernel; two levels
    per lane needs an even nvldim, else the library takes the other one)."""
    K.set_variant(variant)
    K.set_kernel(kernel)
    nE, nC, nV, nA, nvldim = shape
    inp = N.make_inputs(nE, nC, nV, nA, seed=sum(shape[:4]), nvldim=nvldim, ragged=True)
    inp["normalThicknessFlux"][0, :] = -0.0          # sign(1.0, -0.0) = -1 (nested.F90:128)
    ref = N.high_order_flux(inp)
    out = run_hip(K, inp)
    assert np.all(out[nV:, :] == -3.5)               # padding rows are not written
    if variant == 0:
        assert np.array_equal(out[:nV], ref[:nV]), f"max|d|={np.abs(out[:nV] - ref[:nV]).max():.3e}"
    else:
        rel = np.abs(out[:nV] - ref[:nV]) / np.maximum(np.abs(ref[:nV]), 1e-300)
        assert np.all((rel <= 1e-10) | (np.abs(out[:nV] - ref[:nV]) <= 1e-12))   # errTol, nested_vars.F90:36


@pytest.mark.gpu
def test_hip_larger_mesh_ragged_bitwise(K, N):
    """40000 edges, ragged cell counts and level ranges (minLevelCell > 1 on some cells), EXACT: bitwise vs the
    oracle; every kernel choice."""
    K.set_variant(K.VARIANT_EXACT)
    inp = N.make_inputs(40000, 900, 40, 10, seed=77, ragged=True)
    ref = N.high_order_flux(inp)[:40]
    for mode in (2, 1, -1, 0):
        K.set_kernel(mode)
        assert np.array_equal(run_hip(K, inp)[:40], ref), mode
    K.set_kernel(-1)


@pytest.mark.gpu
def test_hip_ignores_out_of_range_cells_and_host_call(K, N):
    K.set_variant(K.VARIANT_EXACT)
    inp = N.make_inputs(50, 12, 20, 4, seed=3)
    ref = N.high_order_flux(inp)
    out = np.full(ref.shape, 9.0, order="F")
    K.high_order_flux_host(inp, out)
    assert np.array_equal(out, ref)
    bad = dict(inp)
    bad["advCellsForEdge"] = inp["advCellsForEdge"].copy(order="F")
    bad["advCellsForEdge"][0, :] = 13                # > nCells: contributes nothing, no fault
    good = dict(inp)
    good["advCoefs"] = inp["advCoefs"].copy(order="F"); good["advCoefs"][0, :] = 0.0
    good["advCoefs3rd"] = inp["advCoefs3rd"].copy(order="F"); good["advCoefs3rd"][0, :] = 0.0
    a, b = run_hip(K, bad), run_hip(K, good)
    assert np.allclose(a, b, rtol=1e-13, atol=1e-9)
