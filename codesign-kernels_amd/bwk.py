"""ctypes binding of libbwk_hip.so (include/bwk_hip.h): the Python mirror of the reference's
`biharmonic_wk_scalar(elem,qtens,deriv,nets,nete)` (atmosphere/biharmonic_wk_kernel.F90:186).

Array convention as in capi.py: the reference's Fortran arrays qtens(np,np,nlev,qsize,nelemd),
Dvv(np,np), elem(144,nelemd) are Fortran-ordered numpy arrays on the host and C-contiguous
torch tensors with reversed axes on the device: qtens (nelemd,qsize,nlev,4,4), dvv (4,4),
elem (nelemd,144).  No CPU fallback: calls raise if the library or a device is missing."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("BWK_HIP_LIB") or os.path.join(HERE, "libbwk_hip.so")
VARIANT_EXACT, VARIANT_FAST = 0, 1
NP, ELEM_DOUBLES = 4, 144
_lib = None


class BwkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbwk_hip error {code}: {msg}")
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise BwkError(-100, f"{_LIB_PATH} not built; run __graft_entry__.build()")
        L = ctypes.CDLL(_LIB_PATH)
        vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.bwk_biharmonic_wk_scalar.restype = ci
        L.bwk_biharmonic_wk_scalar.argtypes = [i64, ci, ci, vp, vp, vp]
        L.bwk_biharmonic_wk_scalar_device.restype = ci
        L.bwk_biharmonic_wk_scalar_device.argtypes = [i64, ci, ci, vp, vp, vp, vp]
        L.bwk_set_variant.restype = ci
        L.bwk_set_variant.argtypes = [ci]
        L.bwk_get_variant.restype = ci
        L.bwk_algorithmic_bytes.restype = i64
        L.bwk_algorithmic_bytes.argtypes = [i64, ci, ci]
        L.bwk_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise BwkError(rc, lib().bwk_last_error().decode())


def set_variant(v):
    return lib().bwk_set_variant(int(v))


def algorithmic_bytes(nelemd, nlev, qsize):
    return int(lib().bwk_algorithmic_bytes(nelemd, nlev, qsize))


def biharmonic_wk_scalar(elem, qtens, dvv, stream=None):
    """Device-resident `call biharmonic_wk_scalar(elem,qtens,deriv,nets,nete)` over all
    elements: torch float64 device tensors qtens (nelemd,qsize,nlev,4,4) -- updated in place --,
    elem (nelemd,144), dvv (4,4); asynchronous on `stream` (default: torch's current stream)."""
    import torch
    for name, t in (("qtens", qtens), ("elem", elem), ("dvv", dvv)):
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise BwkError(-1, f"{name}: need a contiguous float64 device tensor")
    if qtens.dim() != 5 or tuple(qtens.shape[3:]) != (NP, NP):
        raise BwkError(-1, f"qtens: shape {tuple(qtens.shape)} is not (nelemd,qsize,nlev,4,4)")
    nelemd, qsize, nlev = qtens.shape[:3]
    if tuple(elem.shape) != (nelemd, ELEM_DOUBLES) or tuple(dvv.shape) != (NP, NP):
        raise BwkError(-1, f"elem {tuple(elem.shape)} / dvv {tuple(dvv.shape)} do not match qtens")
    s = torch.cuda.current_stream() if stream is None else stream
    _check(lib().bwk_biharmonic_wk_scalar_device(nelemd, nlev, qsize, ctypes.c_void_p(qtens.data_ptr()),
                                                 ctypes.c_void_p(dvv.data_ptr()), ctypes.c_void_p(elem.data_ptr()),
                                                 ctypes.c_void_p(s.cuda_stream)))


def biharmonic_wk_scalar_host(elem, qtens, dvv):
    """The synchronous call on HOST arrays (Fortran-ordered numpy, reference shapes):
    H2D + kernel + D2H."""
    for name, a in (("qtens", qtens), ("elem", elem), ("dvv", dvv)):
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["F_CONTIGUOUS"]):
            raise BwkError(-1, f"{name}: need a Fortran-ordered float64 numpy array")
    if qtens.ndim != 5 or qtens.shape[:2] != (NP, NP):
        raise BwkError(-1, f"qtens: shape {qtens.shape} is not (4,4,nlev,qsize,nelemd)")
    _, _, nlev, qsize, nelemd = qtens.shape
    if elem.shape != (ELEM_DOUBLES, nelemd) or dvv.shape != (NP, NP):
        raise BwkError(-1, "elem / dvv do not match qtens")
    _check(lib().bwk_biharmonic_wk_scalar(nelemd, nlev, qsize, ctypes.c_void_p(qtens.ctypes.data),
                                          ctypes.c_void_p(dvv.ctypes.data), ctypes.c_void_p(elem.ctypes.data)))
