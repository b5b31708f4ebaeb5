"""ctypes binding of libnlk_hip.so (include/nlk_hip.h): the Python mirror of the reference's
high-order flux loop nest (nested_loops/nested.F90:495-559, run_original_cpu_directive).

Device tensors are C-contiguous torch tensors with the reference's axes reversed (level index
last): tracerCur (nCells,nvldim), normalThicknessFlux / advMaskHighOrder / highOrderFlx
(nEdges,nvldim), advCellsForEdge / advCoefs / advCoefs3rd (nEdges,nAdv); int32 index arrays.
No CPU fallback."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("NLK_HIP_LIB") or os.path.join(HERE, "libnlk_hip.so")
VARIANT_EXACT, VARIANT_FAST = 0, 1
INT_KEYS = ("nAdvCellsForEdge", "advCellsForEdge", "minLevelCell", "maxLevelCell")
REAL_KEYS = ("tracerCur", "normalThicknessFlux", "advMaskHighOrder", "advCoefs", "advCoefs3rd")
_lib = None


class NlkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libnlk_hip error {code}: {msg}")
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise NlkError(-100, f"{_LIB_PATH} not built; run __graft_entry__.build()")
        L = ctypes.CDLL(_LIB_PATH)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.nlk_high_order_flux_device.restype = ci
        L.nlk_high_order_flux_device.argtypes = [ci] * 5 + [vp] * 9 + [ctypes.c_double, vp, vp]
        L.nlk_high_order_flux.restype = ci
        L.nlk_high_order_flux.argtypes = [ci] * 5 + [vp] * 9 + [ctypes.c_double, vp]
        L.nlk_set_variant.restype = ci
        L.nlk_set_variant.argtypes = [ci]
        L.nlk_set_kernel.restype = ci
        L.nlk_set_kernel.argtypes = [ci]
        L.nlk_algorithmic_bytes.restype = ctypes.c_int64
        L.nlk_algorithmic_bytes.argtypes = [ci] * 5
        L.nlk_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise NlkError(rc, lib().nlk_last_error().decode())


def set_variant(v):
    return lib().nlk_set_variant(int(v))


def set_kernel(mode):
    """-1 automatic, 0 one level per lane, 1 two levels per lane (nlk_set_kernel); returns the previous setting"""
    return lib().nlk_set_kernel(int(mode))


def algorithmic_bytes(nEdges, nCells, nVertLevels, nvldim, nAdv):
    return int(lib().nlk_algorithmic_bytes(nEdges, nCells, nVertLevels, nvldim, nAdv))


def high_order_flux(d, nVertLevels, coef3rdOrder, highOrderFlx, stream=None):
    """Device-resident call.  `d`: dict of device tensors (INT_KEYS int32, REAL_KEYS float64, axes
    reversed); highOrderFlx (nEdges,nvldim) float64 is written for levels 1..nVertLevels."""
    import torch
    nEdges, nvldim = highOrderFlx.shape
    nCells = d["tracerCur"].shape[0]
    nAdv = d["advCellsForEdge"].shape[1]
    want = {"nAdvCellsForEdge": (nEdges,), "advCellsForEdge": (nEdges, nAdv), "minLevelCell": (nCells,),
            "maxLevelCell": (nCells,), "tracerCur": (nCells, nvldim), "normalThicknessFlux": (nEdges, nvldim),
            "advMaskHighOrder": (nEdges, nvldim), "advCoefs": (nEdges, nAdv), "advCoefs3rd": (nEdges, nAdv)}
    for k, shape in want.items():
        t = d[k]
        dt = torch.int32 if k in INT_KEYS else torch.float64
        if not (t.is_cuda and t.dtype == dt and t.is_contiguous() and tuple(t.shape) == shape):
            raise NlkError(-1, f"{k}: need a contiguous {dt} device tensor of shape {shape}, got {tuple(t.shape)} {t.dtype}")
    if not (highOrderFlx.is_cuda and highOrderFlx.dtype == torch.float64 and highOrderFlx.is_contiguous()):
        raise NlkError(-1, "highOrderFlx: need a contiguous float64 device tensor")
    s = torch.cuda.current_stream() if stream is None else stream
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _check(lib().nlk_high_order_flux_device(nEdges, nCells, int(nVertLevels), nvldim, nAdv,
                                            *[p(d[k]) for k in INT_KEYS], *[p(d[k]) for k in REAL_KEYS],
                                            float(coef3rdOrder), p(highOrderFlx), ctypes.c_void_p(s.cuda_stream)))


def high_order_flux_host(inp, highOrderFlx):
    """Synchronous call on HOST arrays (Fortran-ordered numpy, reference shapes; a dict as
    oracle/nlk.py builds): H2D + kernel + D2H."""
    import numpy as np
    nvldim, nEdges = inp["normalThicknessFlux"].shape
    nCells = inp["tracerCur"].shape[1]
    nAdv = inp["advCellsForEdge"].shape[0]
    ia = [np.asfortranarray(inp[k], dtype=np.int32) for k in INT_KEYS]
    ra = [np.asfortranarray(inp[k], dtype=np.float64) for k in REAL_KEYS]
    if not (highOrderFlx.dtype == np.float64 and highOrderFlx.flags["F_CONTIGUOUS"] and highOrderFlx.shape == (nvldim, nEdges)):
        raise NlkError(-1, "highOrderFlx: need a Fortran-ordered float64 array (nvldim,nEdges)")
    _check(lib().nlk_high_order_flux(nEdges, nCells, int(inp["nVertLevels"]), nvldim, nAdv,
                                     *[ctypes.c_void_p(a.ctypes.data) for a in ia],
                                     *[ctypes.c_void_p(a.ctypes.data) for a in ra],
                                     float(inp["coef3rdOrder"]), ctypes.c_void_p(highOrderFlx.ctypes.data)))
