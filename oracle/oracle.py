"""oracle/oracle.py -- TEST INFRASTRUCTURE: Python face of the CPU oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  It wraps
  * oracle/libmpdata_oracle.so  (mpdata_oracle.c: the C restatement of the
    reference's advect_scalar2D_cpu, reference
    mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:477-642), and
  * oracle/_ref/advect_ref_<shape>  (the reference program itself, built by
    build_ref.py; present only for the shapes that were built),
and provides the synthetic-input generator in numpy (an independent second
implementation of mpdata_oracle_fill, cross-checked in tests).
"""
import ctypes
import os
import re
import resource
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmpdata_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")

# array ids = the reference's fill order (reference :654-660)
SID = {"adz": 0, "f": 1, "u": 2, "w": 3, "rho": 4, "rhow": 5, "flux": 6}
DIST_CONDITIONED, DIST_RAW, DIST_RAW_SIGNED = 1, 2, 3

_lib = None


def build_lib(force=False):
    """gcc the C restatement into oracle/libmpdata_oracle.so."""
    src = os.path.join(HERE, "mpdata_oracle.c")
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= os.path.getmtime(src)):
        return LIB_PATH
    # compiled twice into one library: fp64 symbols, and fp32 symbols (suffix _f32)
    flags = ["-O3", "-ffp-contract=off", "-fopenmp", "-fPIC"]
    with tempfile.TemporaryDirectory(prefix="mpdata_oracle_") as tmp:
        o64, o32 = os.path.join(tmp, "o64.o"), os.path.join(tmp, "o32.o")
        subprocess.run(["gcc", *flags, "-c", "-o", o64, src], check=True)
        subprocess.run(["gcc", *flags, "-DMPDATA_ORACLE_F32", "-c", "-o", o32, src], check=True)
        subprocess.run(["gcc", "-shared", "-fopenmp", "-o", LIB_PATH, o64, o32, "-lm"], check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_lib()
        L = ctypes.CDLL(LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        L.mpdata_oracle_advect.restype = ctypes.c_int
        L.mpdata_oracle_advect.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                           dp, dp, dp, dp, dp, dp, dp, ctypes.c_int]
        L.mpdata_oracle_advect_tracers.restype = ctypes.c_int
        L.mpdata_oracle_advect_tracers.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, dp, dp, dp, dp, dp, dp, dp,
                                                   ctypes.c_int]
        L.mpdata_oracle_max_threads.restype = ctypes.c_int
        L.mpdata_oracle_fill.restype = None
        L.mpdata_oracle_fill.argtypes = [dp, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64,
                                         ctypes.c_int]
        L.mpdata_oracle_advect_stages.restype = ctypes.c_int
        L.mpdata_oracle_advect_stages.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [dp] * 11
        fp = ctypes.POINTER(ctypes.c_float)
        L.mpdata_oracle_advect_f32.restype = ctypes.c_int
        L.mpdata_oracle_advect_f32.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                               fp, fp, fp, fp, fp, fp, fp, ctypes.c_int]
        L.mpdata_oracle_advect_tracers_f32.restype = ctypes.c_int
        L.mpdata_oracle_advect_tracers_f32.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, fp, fp, fp, fp, fp, fp, fp,
                                                       ctypes.c_int]
        L.mpdata_oracle_fill_f32.restype = None
        L.mpdata_oracle_fill_f32.argtypes = [fp, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64,
                                             ctypes.c_int]
        _lib = L
    return _lib


def _dp(a):
    """ctypes pointer to a Fortran-ordered fp64 or fp32 array."""
    assert a.dtype in (np.float64, np.float32) and a.flags["F_CONTIGUOUS"]
    ct = ctypes.c_double if a.dtype == np.float64 else ctypes.c_float
    return a.ctypes.data_as(ctypes.POINTER(ct))


def shapes(ncrms, nx, nz, ntracers=1):
    """Fortran-order array shapes (reference :30-38); the singleton j axis is
    dropped, the tracer axis (slowest) is this build's extension."""
    nzm = nz - 1
    sh = {"adz": (ncrms, nzm), "f": (ncrms, nx + 6, nzm), "u": (ncrms, nx + 5, nzm),
          "w": (ncrms, nx + 4, nz), "rho": (ncrms, nzm), "rhow": (ncrms, nz),
          "flux": (ncrms, nz)}
    if ntracers > 1:
        sh["f"] = sh["f"] + (ntracers,)
        sh["flux"] = sh["flux"] + (ntracers,)
    return sh


_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def fill_array(name, shape, seed, dist, ncrms_global=None, sl0=0, dtype=np.float64):
    """numpy version of mpdata_oracle_fill: one input array, Fortran order.
    shape[0] is the local ncrms; ncrms_global/sl0 place it inside a larger
    (sharded) problem.  dtype float32 rounds the fp64 value (reference precision
    switch :11-12)."""
    nloc = shape[0]
    rows = int(np.prod(shape[1:], dtype=np.int64))
    ng = nloc if ncrms_global is None else ncrms_global
    sid = SID[name]
    with np.errstate(over="ignore"):
        base = np.uint64(seed) + np.uint64(sid) * np.uint64(0xD1B54A32D192ED03)
        r = np.arange(rows, dtype=np.uint64)[None, :] * np.uint64(ng)
        s = np.arange(nloc, dtype=np.uint64)[:, None] + np.uint64(sl0)
        j = r + s
        z = _mix64(base + (j + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
    out = (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    shift = 0.0
    if dist == DIST_CONDITIONED:
        shift = {2: -0.5, 3: -0.5, 0: 0.5, 4: 0.5, 5: 0.5}.get(sid, 0.0)
    elif dist == DIST_RAW_SIGNED:
        shift = {2: -0.5, 3: -0.5}.get(sid, 0.0)
    out = (out + shift).astype(dtype)
    return np.asfortranarray(out.reshape(shape, order="F"))


def make_inputs(ncrms, nx, nz, seed=100, dist=DIST_CONDITIONED, ntracers=1,
                ncrms_global=None, sl0=0, dtype=np.float64):
    """All seven arrays as a dict of Fortran-ordered float64 (or float32) arrays."""
    sh = shapes(ncrms, nx, nz, ntracers)
    return {k: fill_array(k, sh[k], seed, dist, ncrms_global, sl0, dtype) for k in SID}


def fill_array_c(name, shape, seed, dist, ncrms_global=None, sl0=0, dtype=np.float64):
    """The C generator (mpdata_oracle_fill[_f32]); must equal fill_array bitwise."""
    nloc = shape[0]
    rows = int(np.prod(shape[1:], dtype=np.int64))
    a = np.empty(shape, dtype=dtype, order="F")
    fn = lib().mpdata_oracle_fill if a.dtype == np.float64 else lib().mpdata_oracle_fill_f32
    fn(_dp(a), SID[name], rows, nloc if ncrms_global is None else ncrms_global, sl0, nloc,
       seed, dist)
    return a


def advect(inp, nthreads=1):
    """Run the C restatement on a dict from make_inputs (not modified).
    Returns (f_out, flux_out)."""
    f = np.array(inp["f"], order="F", copy=True)
    flux = np.array(inp["flux"], order="F", copy=True)
    ncrms, nxp6, nzm = f.shape[:3]
    nt = f.shape[3] if f.ndim == 4 else 1
    nx, nz = nxp6 - 6, nzm + 1
    args = (_dp(f), _dp(inp["u"]), _dp(inp["w"]), _dp(inp["rho"]), _dp(inp["rhow"]),
            _dp(inp["adz"]), _dp(flux), nthreads)
    sfx = "" if f.dtype == np.float64 else "_f32"
    if nt == 1:
        rc = getattr(lib(), "mpdata_oracle_advect" + sfx)(ncrms, nx, nz, *args)
    else:
        rc = getattr(lib(), "mpdata_oracle_advect_tracers" + sfx)(ncrms, nx, nz, nt, *args)
    if rc != 0:
        raise RuntimeError(f"mpdata_oracle_advect failed rc={rc}")
    return f, flux


def stage_shapes(ncrms, nx, nz):
    """Fortran-order shapes of the reference's temporaries (reference :485-491)."""
    nzm = nz - 1
    return {"uuu": (ncrms, nx + 5, nzm), "www": (ncrms, nx + 4, nz), "mx": (ncrms, nx + 2, nzm),
            "mn": (ncrms, nx + 2, nzm)}


def advect_stages(inp, last_stage, fill=0.0):
    """The routine stopped after stage `last_stage` (1..8, see mpdata_oracle_advect_stages);
    returns a dict with f, flux and the temporaries uuu, www, mx, mn (fp64 only).  The
    temporaries start as `fill` (parts of them are never written, as in the reference)."""
    f = np.array(inp["f"], order="F", copy=True)
    flux = np.array(inp["flux"], order="F", copy=True)
    ncrms, nxp6, nzm = f.shape
    nx, nz = nxp6 - 6, nzm + 1
    tmp = {k: np.full(s, fill, dtype=np.float64, order="F") for k, s in stage_shapes(ncrms, nx, nz).items()}
    rc = lib().mpdata_oracle_advect_stages(ncrms, nx, nz, last_stage, _dp(f), _dp(inp["u"]), _dp(inp["w"]),
                                           _dp(inp["rho"]), _dp(inp["rhow"]), _dp(inp["adz"]), _dp(flux),
                                           _dp(tmp["uuu"]), _dp(tmp["www"]), _dp(tmp["mx"]), _dp(tmp["mn"]))
    if rc != 0:
        raise RuntimeError(f"mpdata_oracle_advect_stages failed rc={rc}")
    return {"f": f, "flux": flux, **tmp}


def max_threads():
    return lib().mpdata_oracle_max_threads()


# ---------------------------------------------------------------- reference
def ref_exe(ncrms, nx, nz, dtype=np.float64):
    sfx = "" if np.dtype(dtype) == np.float64 else "_f32"
    p = os.path.join(REF_DIR, f"advect_ref_{ncrms}x{nx}x{nz}{sfx}")
    return p if os.path.exists(p) else None


def _unlimit_stack():
    # the reference keeps its temporaries in automatic arrays (reference
    # :485-491); ncrms=4096 overflows the default 8 MB stack
    resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))


def run_reference(inp, want_outputs=True):
    """Run the reference executable for this shape (single tracer) on `inp`.
    Returns (f_out, flux_out, cpu_timing_seconds)."""
    ncrms, nxp6, nzm = inp["f"].shape
    nx, nz = nxp6 - 6, nzm + 1
    dtype = inp["f"].dtype
    exe = ref_exe(ncrms, nx, nz, dtype)
    if exe is None:
        raise FileNotFoundError(f"no oracle/_ref binary for shape {(ncrms, nx, nz)}")
    with tempfile.TemporaryDirectory(prefix="mpdata_refrun_") as tmp:
        with open(os.path.join(tmp, "mpdata_in.bin"), "wb") as fh:
            for k in ("adz", "f", "u", "w", "rho", "rhow", "flux"):
                fh.write(np.asfortranarray(inp[k]).tobytes(order="F"))
        res = subprocess.run([exe], cwd=tmp, check=True, capture_output=True, text=True,
                             preexec_fn=_unlimit_stack)
        m = re.search(r"CPU Timing:\s*([0-9.Ee+-]+)", res.stdout)
        timing = float(m.group(1)) if m else float("nan")
        if not want_outputs:
            return None, None, timing
        raw = np.fromfile(os.path.join(tmp, "mpdata_out.bin"), dtype=dtype)
    nf = inp["f"].size
    f = raw[:nf].reshape(inp["f"].shape, order="F")
    flux = raw[nf:nf + inp["flux"].size].reshape(inp["flux"].shape, order="F")
    return np.asfortranarray(f), np.asfortranarray(flux), timing


def rel_l1(a, b):
    """The reference's own metric (reference :681-682): sum|a-b| / sum|b|."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.sum(np.abs(a - b)) / np.sum(np.abs(b)))
