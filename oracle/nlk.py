"""oracle/nlk.py -- TEST INFRASTRUCTURE: Python face of the CPU oracle of the third mini-app
(nested_loops/nested.F90: MPAS-Ocean high-order tracer flux gather).  Wraps
oracle/libnlk_oracle.so (nlk_oracle.c) and oracle/_ref/nlk_ref (the reference program itself,
built by build_ref.py --nlk).  Only tests/, smoke() and bench.py's cpu_baseline leg import it."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnlk_oracle.so")
REF_EXE = os.path.join(HERE, "_ref", "nlk_ref")
INT_KEYS = ("nAdvCellsForEdge", "advCellsForEdge", "minLevelCell", "maxLevelCell")
REAL_KEYS = ("tracerCur", "normalThicknessFlux", "advMaskHighOrder", "advCoefs", "advCoefs3rd")
_lib = None


def build_lib(force=False):
    src = os.path.join(HERE, "nlk_oracle.c")
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= os.path.getmtime(src):
        return LIB_PATH
    subprocess.run(["gcc", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB_PATH, src, "-lm"], check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_lib()
        L = ctypes.CDLL(LIB_PATH)
        ip, dp, ci = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double), ctypes.c_int
        L.nlk_oracle_high_order_flux.restype = ci
        L.nlk_oracle_high_order_flux.argtypes = [ci] * 5 + [ip] * 4 + [dp] * 5 + [ctypes.c_double, dp]
        L.nlk_oracle_coef3rd.restype = ctypes.c_double
        _lib = L
    return _lib


def coef3rd():
    """coef3rdOrder as the reference holds it (nested_vars.F90:35: the fp32 value of 2.14)."""
    return lib().nlk_oracle_coef3rd()


def make_inputs(nEdges, nCells, nVertLevels, nAdv, seed, nvldim=None, ragged=True):
    """Inputs in the spirit of the reference's initialisation (nested.F90:58-108): random
    connectivity, topography with about half the cells at full depth; `ragged` also varies
    nAdvCellsForEdge and minLevelCell (the reference keeps them at nAdv and 1)."""
    rng = np.random.default_rng(seed)
    nvldim = nVertLevels if nvldim is None else nvldim
    maxl = np.minimum(np.maximum(3, np.rint(rng.random(nCells) * nVertLevels * 2.0)), nVertLevels).astype(np.int32)
    maxl = np.minimum(maxl, nVertLevels)
    minl = np.ones(nCells, np.int32)
    if ragged:
        minl = np.minimum(rng.integers(1, 4, nCells), maxl).astype(np.int32)
    tr = np.zeros((nvldim, nCells), order="F")
    for c in range(nCells):
        tr[minl[c] - 1:maxl[c], c] = 15.0 * rng.random(maxl[c] - minl[c] + 1)
    nadv = np.full(nEdges, nAdv, np.int32)
    if ragged:
        nadv = rng.integers(1, nAdv + 1, nEdges).astype(np.int32)
    inp = {"nAdvCellsForEdge": nadv,
           "advCellsForEdge": np.asfortranarray(rng.integers(1, nCells + 1, (nAdv, nEdges)).astype(np.int32)),
           "minLevelCell": minl, "maxLevelCell": maxl, "tracerCur": tr,
           "normalThicknessFlux": np.asfortranarray(15.0 * (0.5 - rng.random((nvldim, nEdges)))),
           "advMaskHighOrder": np.asfortranarray((rng.random((nvldim, nEdges)) < 0.9).astype(np.float64) if ragged
                                                 else np.ones((nvldim, nEdges))),
           "advCoefs": np.asfortranarray(20.0 * rng.random((nAdv, nEdges))),
           "advCoefs3rd": np.asfortranarray(21.0 * rng.random((nAdv, nEdges))),
           "coef3rdOrder": coef3rd(), "nVertLevels": nVertLevels}
    return inp


def high_order_flux(inp):
    """C restatement of the reference loop (nested.F90:123-157) -> highOrderFlx(nvldim,nEdges);
    rows nVertLevels+1..nvldim stay 0."""
    nvldim, nEdges = inp["normalThicknessFlux"].shape
    nCells = inp["tracerCur"].shape[1]
    nAdv = inp["advCellsForEdge"].shape[0]
    out = np.zeros((nvldim, nEdges), order="F")
    ip, dp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
    ia = [np.asfortranarray(inp[k], dtype=np.int32) for k in INT_KEYS]
    ra = [np.asfortranarray(inp[k], dtype=np.float64) for k in REAL_KEYS]
    rc = lib().nlk_oracle_high_order_flux(nEdges, nCells, inp["nVertLevels"], nvldim, nAdv,
                                          *[a.ctypes.data_as(ip) for a in ia], *[a.ctypes.data_as(dp) for a in ra],
                                          float(inp["coef3rdOrder"]), out.ctypes.data_as(dp))
    if rc:
        raise RuntimeError(f"nlk_oracle_high_order_flux failed rc={rc}")
    return out


def run_reference(nEdges, nCells, nVertLevels, nAdv, seed):
    """Run the reference program on a seeded random stream; returns (inputs as the program built
    them, its refFlx)."""
    if not os.path.exists(REF_EXE):
        raise FileNotFoundError("no oracle/_ref/nlk_ref")
    nrand = nCells * (1 + nVertLevels) + nEdges * (3 * nAdv + nVertLevels) + 16
    with tempfile.TemporaryDirectory(prefix="nlk_refrun_") as tmp:
        with open(os.path.join(tmp, "nested.nml"), "w") as fh:
            fh.write(f"&nested_nml\n nIters = 1\n nEdges = {nEdges}\n nCells = {nCells}\n"
                     f" nVertLevels = {nVertLevels}\n nAdv = {nAdv}\n/\n")
        np.random.default_rng(seed).random(nrand).tofile(os.path.join(tmp, "nlk_rand.bin"))
        subprocess.run([REF_EXE], cwd=tmp, check=True, capture_output=True, text=True)
        raw = open(os.path.join(tmp, "nlk_out.bin"), "rb").read()
    off = 0

    def take(dtype, shape):
        nonlocal off
        n = int(np.prod(shape))
        a = np.frombuffer(raw, dtype=dtype, count=n, offset=off).reshape(shape, order="F").copy(order="F")
        off += n * np.dtype(dtype).itemsize
        return a

    nE, nC, nV, nvldim, nA = (int(x) for x in take(np.int32, (5,)))
    inp = {"nAdvCellsForEdge": take(np.int32, (nE,)), "advCellsForEdge": take(np.int32, (nA, nE)),
           "minLevelCell": take(np.int32, (nC,)), "maxLevelCell": take(np.int32, (nC,)),
           "tracerCur": take(np.float64, (nvldim, nC)), "normalThicknessFlux": take(np.float64, (nvldim, nE)),
           "advMaskHighOrder": take(np.float64, (nvldim, nE)), "advCoefs": take(np.float64, (nA, nE)),
           "advCoefs3rd": take(np.float64, (nA, nE))}
    inp["coef3rdOrder"] = float(take(np.float64, (1,))[0])
    inp["nVertLevels"] = nV
    ref = take(np.float64, (nvldim, nE))
    return inp, ref
