"""oracle/bwk.py -- TEST INFRASTRUCTURE: Python face of the CPU oracle of the second
mini-app (atmosphere/biharmonic_wk_kernel.F90).  Wraps
oracle/libbwk_oracle.so (bwk_oracle.c) and oracle/_ref/bwk_ref_ne<N> (the reference program
itself, built by build_ref.py --bwk).  Only tests/, smoke() and bench.py's cpu_baseline leg
import this module."""
import ctypes
import os
import re
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbwk_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")
NP, NLEV, QSIZE = 4, 72, 40      # reference :8-10
ELEM_DOUBLES = 144               # Dinv(4,4,2,2) | spheremp(4,4) | tensorVisc(4,4,2,2), reference :23-27

_lib = None


def build_lib(force=False):
    src = os.path.join(HERE, "bwk_oracle.c")
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= os.path.getmtime(src):
        return LIB_PATH
    subprocess.run(["gcc", "-O3", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", LIB_PATH, src, "-lm"],
                   check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_lib()
        L = ctypes.CDLL(LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        L.bwk_oracle_biharmonic.restype = ctypes.c_int
        L.bwk_oracle_biharmonic.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, dp, dp, dp, ctypes.c_int]
        L.bwk_oracle_init.restype = None
        L.bwk_oracle_init.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, dp, dp, dp]
        _lib = L
    return _lib


def _dp(a):
    assert a.dtype == np.float64 and a.flags["F_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def make_inputs(nelemd, nlev=NLEV, qsize=QSIZE):
    """The reference's own inputs (initialize_data, reference :48-58): dict of Fortran-ordered
    arrays dvv(4,4), elem(144,nelemd), qtens(4,4,nlev,qsize,nelemd)."""
    dvv = np.zeros((NP, NP), order="F")
    elem = np.zeros((ELEM_DOUBLES, nelemd), order="F")
    qtens = np.zeros((NP, NP, nlev, qsize, nelemd), order="F")
    lib().bwk_oracle_init(nelemd, nlev, qsize, _dp(dvv), _dp(elem), _dp(qtens))
    return {"dvv": dvv, "elem": elem, "qtens": qtens}


def random_inputs(nelemd, nlev, qsize, seed):
    """Other data than the reference's LCG (sizes and values the reference never runs)."""
    rng = np.random.default_rng(seed)
    return {"dvv": np.asfortranarray(rng.uniform(-1, 1, (NP, NP))),
            "elem": np.asfortranarray(rng.uniform(-1, 1, (ELEM_DOUBLES, nelemd))),
            "qtens": np.asfortranarray(rng.uniform(-1, 1, (NP, NP, nlev, qsize, nelemd)))}


def biharmonic(inp, nthreads=1):
    """C restatement of biharmonic_wk_scalar (reference :186-200) on a copy of inp['qtens']."""
    q = np.array(inp["qtens"], order="F", copy=True)
    _, _, nlev, qsize, nelemd = q.shape
    rc = lib().bwk_oracle_biharmonic(nelemd, nlev, qsize, _dp(inp["dvv"]), _dp(inp["elem"]), _dp(q), nthreads)
    if rc:
        raise RuntimeError("bwk_oracle_biharmonic failed")
    return q


def ref_exe(nelemd):
    p = os.path.join(REF_DIR, f"bwk_ref_ne{nelemd}")
    return p if os.path.exists(p) else None


def run_reference(nelemd):
    """Run the reference executable; returns (inputs dict as the program generated them, the
    CPU routine's qtens, its 'CPU time' seconds)."""
    exe = ref_exe(nelemd)
    if exe is None:
        raise FileNotFoundError(f"no oracle/_ref/bwk_ref_ne{nelemd}")
    with tempfile.TemporaryDirectory(prefix="bwk_refrun_") as tmp:
        res = subprocess.run([exe], cwd=tmp, check=True, capture_output=True, text=True)
        m = re.search(r"CPU\s+time:\s*([0-9.Ee+-]+)", res.stdout)
        t = float(m.group(1)) if m else float("nan")
        raw = np.fromfile(os.path.join(tmp, "bwk_in.bin"), dtype=np.float64)
        out = np.fromfile(os.path.join(tmp, "bwk_out.bin"), dtype=np.float64)
    shp = (NP, NP, NLEV, QSIZE, nelemd)
    n_el = ELEM_DOUBLES * nelemd
    inp = {"dvv": raw[:16].reshape((NP, NP), order="F").copy(order="F"),
           "elem": raw[16:16 + n_el].reshape((ELEM_DOUBLES, nelemd), order="F").copy(order="F"),
           "qtens": raw[16 + n_el:].reshape(shp, order="F").copy(order="F")}
    return inp, out.reshape(shp, order="F").copy(order="F"), t


def l2norm(a, b):
    """The reference's own metric (reference :69-73)."""
    return float(np.sqrt(np.sum((a - b) ** 2) / np.sum(b ** 2)))
