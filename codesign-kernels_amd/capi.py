"""ctypes binding of libmpdata_hip.so (include/mpdata_hip.h) and the Python
mirror of the reference's operator interface.

Array convention.  The reference declares (Fortran, column-major, `sl`
fastest; reference :479-484, :30)
    f(ncrms,-2:nx+3,1,nzm)  u(ncrms,-1:nx+3,1,nzm)  w(ncrms,-1:nx+2,1,nz)
    rho(ncrms,nzm)  rhow(ncrms,nz)  adz(ncrms,nzm)  flux(ncrms,nz)
The same bytes are
  * numpy: Fortran-ordered arrays of shape (ncrms, nx+6, nzm[, ntracers]) ...
  * torch: C-contiguous tensors with the axes reversed,
           ([ntracers,] nzm, nx+6, ncrms) ...
so a device tensor's last axis is the coalesced CRM-instance axis.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MPDATA_HIP_LIB: load an alternative build of the library (kernel experiments)
_LIB_PATH = os.environ.get("MPDATA_HIP_LIB") or os.path.join(HERE, "libmpdata_hip.so")
VARIANT_EXACT, VARIANT_FAST = 0, 1
EINVAL, EUNSUPPORTED, ESTATE, ECOMM = -1, -2, -3, -4   # MPDATA_E* of include/mpdata_hip.h
LAYOUT_REFERENCE, LAYOUT_WAVEMAJOR = 0, 1

_lib = None


class MpdataError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmpdata_hip error {code}: {msg}")
        self.code = code


def lib_path():
    return _LIB_PATH


def build_library(force=False):
    """hipcc --offload-arch=gfx950 the kernels + C-ABI into libmpdata_hip.so
    (cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(HERE, "csrc"), "-j4"]
    if force:
        args.append("-B")
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise MpdataError(-100, f"{_LIB_PATH} not built; run __graft_entry__.build() "
                                    "(make -C codesign-kernels_amd/csrc)")
        L = ctypes.CDLL(_LIB_PATH)
        dp, vp = ctypes.c_void_p, ctypes.c_void_p
        i64, ci = ctypes.c_int64, ctypes.c_int
        L.mpdata_advect_scalar2d.restype = ci
        L.mpdata_advect_scalar2d.argtypes = [i64, ci, ci, ci] + [dp] * 7
        L.mpdata_advect_scalar2d_device.restype = ci
        L.mpdata_advect_scalar2d_device.argtypes = [i64, ci, ci, ci] + [dp] * 7 + [vp]
        L.mpdata_plan_create.restype = ci
        L.mpdata_plan_create.argtypes = [i64, ci, ci, ci, ctypes.POINTER(vp)]
        L.mpdata_plan_upload.restype = ci
        L.mpdata_plan_upload.argtypes = [vp] + [dp] * 7
        for name in ("mpdata_plan_run", "mpdata_plan_sync", "mpdata_plan_destroy"):
            getattr(L, name).restype = ci
            getattr(L, name).argtypes = [vp]
        L.mpdata_plan_download.restype = ci
        L.mpdata_plan_download.argtypes = [vp, dp, dp]
        L.mpdata_plan_last_kernel_ms.restype = ci
        L.mpdata_plan_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.mpdata_plan_run_tracers.restype = ci
        L.mpdata_plan_run_tracers.argtypes = [vp, ci, ci]
        L.mpdata_plan_set_timing.restype = ci
        L.mpdata_plan_set_timing.argtypes = [vp, ci]
        L.mpdata_plan_run_uw.restype = ci
        L.mpdata_plan_run_uw.argtypes = [vp, ci, ci, vp, vp]
        L.mpdata_plan_import_device.restype = ci
        L.mpdata_plan_import_device.argtypes = [vp] + [vp] * 7 + [ci, ci]
        L.mpdata_plan_export_device.restype = ci
        L.mpdata_plan_export_device.argtypes = [vp, vp, vp, ci, ci]
        L.mpdata_plan_set_stream.restype = ci
        L.mpdata_plan_set_stream.argtypes = [vp, vp]
        for name in ("mpdata_plan_layout", "mpdata_plan_device"):
            getattr(L, name).restype = ci
            getattr(L, name).argtypes = [vp]
        L.mpdata_set_plan_layout.restype = ci
        L.mpdata_set_plan_layout.argtypes = [ci]
        L.mpdata_plan_create_multi.restype = ci
        L.mpdata_plan_create_multi.argtypes = [i64, ci, ci, ci, ci, ctypes.POINTER(vp)]
        L.mpdata_plan_create_multi_devices.restype = ci
        L.mpdata_plan_create_multi_devices.argtypes = [i64, ci, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(vp)]
        L.mpdata_shard_range.restype = None
        L.mpdata_shard_range.argtypes = [i64, ci, ci, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.mpdata_plan_ngpus.restype = ci
        L.mpdata_plan_ngpus.argtypes = [vp]
        L.mpdata_plan_ranks_seen.restype = ci
        L.mpdata_plan_ranks_seen.argtypes = [vp]
        L.mpdata_plan_shard.restype = ci
        L.mpdata_plan_shard.argtypes = [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(i64), ctypes.POINTER(i64)]
        L.mpdata_plan_shard_plan.restype = vp
        L.mpdata_plan_shard_plan.argtypes = [vp, ci]
        L.mpdata_plan_transfer_stats.restype = ci
        L.mpdata_plan_transfer_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                                 ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(ci)]
        L.mpdata_fill_synthetic_device.restype = ci
        L.mpdata_fill_synthetic_device.argtypes = [dp, ci, i64, i64, i64, i64, ctypes.c_uint64, ci, vp]
        L.mpdata_pack_shard_device.restype = ci
        L.mpdata_pack_shard_device.argtypes = [dp, dp, i64, i64, i64, i64, vp]
        L.mpdata_unpack_shard_device.restype = ci
        L.mpdata_unpack_shard_device.argtypes = [dp, dp, i64, i64, i64, i64, vp]
        L.mpdata_advect_scalar2d_f32.restype = ci
        L.mpdata_advect_scalar2d_f32.argtypes = [i64, ci, ci, ci] + [dp] * 7
        L.mpdata_advect_scalar2d_f32_device.restype = ci
        L.mpdata_advect_scalar2d_f32_device.argtypes = [i64, ci, ci, ci] + [dp] * 7 + [vp]
        L.mpdata_fill_synthetic_f32_device.restype = ci
        L.mpdata_fill_synthetic_f32_device.argtypes = [dp, ci, i64, i64, i64, i64, ctypes.c_uint64, ci, vp]
        L.mpdata_algorithmic_bytes_f32.restype = i64
        L.mpdata_algorithmic_bytes_f32.argtypes = [i64, ci, ci, ci]
        L.mpdata_plan_create_f32.restype = ci
        L.mpdata_plan_create_f32.argtypes = [i64, ci, ci, ci, ctypes.POINTER(vp)]
        L.mpdata_plan_upload_f32.restype = ci
        L.mpdata_plan_upload_f32.argtypes = [vp] + [dp] * 7
        L.mpdata_plan_download_f32.restype = ci
        L.mpdata_plan_download_f32.argtypes = [vp, dp, dp]
        L.mpdata_debug_stages_device.restype = ci
        L.mpdata_debug_stages_device.argtypes = [i64, ci, ci, ci] + [dp] * 11 + [vp]
        L.mpdata_set_variant.restype = ci
        L.mpdata_set_variant.argtypes = [ci]
        L.mpdata_get_variant.restype = ci
        L.mpdata_set_serpentine.restype = ci
        L.mpdata_set_serpentine.argtypes = [ci]
        L.mpdata_set_wm_flags.restype = ci
        L.mpdata_set_wm_flags.argtypes = [ci]
        L.mpdata_set_tile.restype = ci
        L.mpdata_set_tile.argtypes = [ci]
        L.mpdata_set_debug_buffer.restype = ci
        L.mpdata_set_debug_buffer.argtypes = [vp]
        L.mpdata_device_count.restype = ci
        L.mpdata_plan_device_alloc.restype = ci
        L.mpdata_plan_device_alloc.argtypes = [vp, ctypes.POINTER(vp), i64]
        L.mpdata_device_free.restype = ci
        L.mpdata_device_free.argtypes = [vp]
        L.mpdata_algorithmic_bytes.restype = i64
        L.mpdata_algorithmic_bytes.argtypes = [i64, ci, ci, ci]
        L.mpdata_diag_stream_3r1w.restype = ci
        L.mpdata_diag_stream_3r1w.argtypes = [i64, ci, ci, ctypes.POINTER(ctypes.c_double)]
        L.mpdata_last_error.restype = ctypes.c_char_p
        L.mpdata_version.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise MpdataError(rc, lib().mpdata_last_error().decode())


def set_variant(v):
    return lib().mpdata_set_variant(int(v))


def get_variant():
    return lib().mpdata_get_variant()


WMF_NOSTREAM, WMF_TPW1, WMF_NOSPLIT, WMF_SPLIT = 1, 2, 4, 8


def set_serpentine(on):
    """Serpentine tile order of wave-major plans (include/mpdata_hip.h section 8); returns the previous setting."""
    return lib().mpdata_set_serpentine(int(on))


def set_wm_flags(flags):
    """Test switches of the wave-major launch (WMF_*; < 0 queries); returns the previous value."""
    return lib().mpdata_set_wm_flags(int(flags))


def version():
    return lib().mpdata_version().decode()


def set_tile(t):
    return lib().mpdata_set_tile(int(t))


def set_plan_layout(layout):
    """Default device layout of new plans (LAYOUT_*); returns the previous one."""
    return lib().mpdata_set_plan_layout(int(layout))


def shard_range(ncrms, ngpus, g):
    """(sl0, nloc) of GPU g's contiguous block of CRM instances (mpdata_shard_range)."""
    a, b = ctypes.c_int64(), ctypes.c_int64()
    lib().mpdata_shard_range(int(ncrms), int(ngpus), int(g), ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def stream_ceiling(bytes_per_array=538 * 2**20, nontemporal=True, iters=40):
    """GB/s of a linear 3-read-1-write stream on the current device (mpdata_diag_stream_3r1w)."""
    g = ctypes.c_double()
    _check(lib().mpdata_diag_stream_3r1w(int(bytes_per_array), int(bool(nontemporal)), int(iters), ctypes.byref(g)))
    return g.value


def device_count():
    return lib().mpdata_device_count()


def algorithmic_bytes(ncrms, nx, nz, ntracers=1, f32=False):
    fn = lib().mpdata_algorithmic_bytes_f32 if f32 else lib().mpdata_algorithmic_bytes
    return int(fn(ncrms, nx, nz, ntracers))


SID = {"adz": 0, "f": 1, "u": 2, "w": 3, "rho": 4, "rhow": 5, "flux": 6}


def shapes(ncrms, nx, nz, ntracers=1):
    """torch-side (C-order, reversed axes) shapes of the seven arrays."""
    nzm = nz - 1
    t = (ntracers,) if ntracers > 1 else ()
    return {"adz": (nzm, ncrms), "f": t + (nzm, nx + 6, ncrms), "u": (nzm, nx + 5, ncrms),
            "w": (nz, nx + 4, ncrms), "rho": (nzm, ncrms), "rhow": (nz, ncrms),
            "flux": t + (nz, ncrms)}


# Placement advice for callers that own the device arrays: a
# workgroup reads the same instance range of f, u and w at about the same time; when the
# three base addresses are equal modulo 1 KiB those requests meet on the same HBM channel
# (8 % slower at ncrms = 65536).  The library's own buffers (plans, host-array calls) are
# staggered the same way.
STAGGER_BYTES = {"f": 0, "u": 256, "w": 512, "rho": 768}


def empty_staggered(shape, name, dtype=None, device="cuda"):
    """Uninitialised device tensor for array `name` whose base address is
    STAGGER_BYTES[name] modulo 1 KiB (a view into a slightly larger allocation)."""
    import torch
    dtype = torch.float64 if dtype is None else dtype
    n = 1
    for x in shape:
        n *= int(x)
    isz = torch.empty((), dtype=dtype).element_size()
    raw = torch.empty(n + 2048 // isz, dtype=dtype, device=device)
    off = (STAGGER_BYTES.get(name, 0) - raw.data_ptr()) % 1024
    return raw[off // isz: off // isz + n].view(tuple(shape))


def _stream_handle(stream):
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return ctypes.c_void_p(s.cuda_stream)


def _dims_from(f, u):
    """(ncrms, nx, nz, ntracers) from torch tensors in the reversed-axes layout."""
    nt = f.shape[0] if f.dim() == 4 else 1
    nzm, nxp6, ncrms = f.shape[-3:]
    if tuple(u.shape) != (nzm, nxp6 - 1, ncrms):
        raise MpdataError(-1, f"u shape {tuple(u.shape)} does not match f {tuple(f.shape)}")
    return ncrms, nxp6 - 6, nzm + 1, nt


def _dev_ptr(t, shape, name, dtype=None):
    import torch
    dtype = torch.float64 if dtype is None else dtype
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise MpdataError(-1, f"{name}: need a contiguous {dtype} device tensor")
    if tuple(t.shape) != tuple(shape):
        raise MpdataError(-1, f"{name}: shape {tuple(t.shape)} != expected {tuple(shape)}")
    return ctypes.c_void_p(t.data_ptr())


def advect_scalar2D(f, u, w, rho, rhow, flux, adz, stream=None):
    """Device-resident `call advect_scalar2D(f,u,w,rho,rhow,flux)` (reference
    :53/:57; adz is host-associated there, :30).  torch float64 device tensors
    (or float32, all seven: the reference's `rp` switch, :12-13) in the
    reversed-axes layout; f and flux are updated in place; asynchronous on
    `stream` (default: torch's current stream)."""
    import torch
    ncrms, nx, nz, nt = _dims_from(f, u)
    sh = shapes(ncrms, nx, nz, nt)
    if f.dtype not in (torch.float64, torch.float32):
        raise MpdataError(-1, f"f: dtype {f.dtype} is neither float64 nor float32")
    ptrs = [_dev_ptr(t, sh[k], k, f.dtype) for k, t in
            (("f", f), ("u", u), ("w", w), ("rho", rho), ("rhow", rhow), ("adz", adz), ("flux", flux))]
    fn = lib().mpdata_advect_scalar2d_device if f.dtype == torch.float64 else lib().mpdata_advect_scalar2d_f32_device
    _check(fn(ncrms, nx, nz, nt, *ptrs, _stream_handle(stream)))


def stage_shapes(ncrms, nx, nz):
    """torch-side shapes of the reference's temporaries (reference :485-491)."""
    nzm = nz - 1
    return {"uuu": (nzm, nx + 5, ncrms), "www": (nz, nx + 4, ncrms), "mx": (nzm, nx + 2, ncrms),
            "mn": (nzm, nx + 2, ncrms)}


def debug_stages(f, u, w, rho, rhow, flux, adz, tmp, last_stage, stream=None):
    """Stage-by-stage debug mode (include/mpdata_hip.h section 7): run stages 1..last_stage
    unfused; `tmp` = dict of float64 device tensors uuu, www, mx, mn (stage_shapes)."""
    import torch
    ncrms, nx, nz, nt = _dims_from(f, u)
    if nt != 1:
        raise MpdataError(-1, "debug_stages: one tracer")
    sh = shapes(ncrms, nx, nz, 1)
    ptrs = [_dev_ptr(t, sh[k], k, torch.float64) for k, t in
            (("f", f), ("u", u), ("w", w), ("rho", rho), ("rhow", rhow), ("adz", adz), ("flux", flux))]
    ssh = stage_shapes(ncrms, nx, nz)
    tptrs = [_dev_ptr(tmp[k], ssh[k], k, torch.float64) for k in ("uuu", "www", "mx", "mn")]
    _check(lib().mpdata_debug_stages_device(ncrms, nx, nz, int(last_stage), *ptrs, *tptrs,
                                            _stream_handle(stream)))


def _host_ptr(a, name, writable=False, dtype=np.float64):
    if not (isinstance(a, np.ndarray) and a.dtype == dtype and a.flags["F_CONTIGUOUS"]):
        raise MpdataError(-1, f"{name}: need a Fortran-ordered {np.dtype(dtype).name} numpy array")
    if writable and not a.flags["WRITEABLE"]:
        raise MpdataError(-1, f"{name}: not writable")
    return ctypes.c_void_p(a.ctypes.data)


def _host_dims(f):
    ncrms, nxp6, nzm = f.shape[:3]
    nt = f.shape[3] if f.ndim == 4 else 1
    return ncrms, nxp6 - 6, nzm + 1, nt


def host_shapes(ncrms, nx, nz, ntracers=1):
    """numpy-side (Fortran order, the reference's declarations :479-484, :30) shapes."""
    nzm = nz - 1
    t = (ntracers,) if ntracers > 1 else ()
    return {"adz": (ncrms, nzm), "f": (ncrms, nx + 6, nzm) + t, "u": (ncrms, nx + 5, nzm),
            "w": (ncrms, nx + 4, nz), "rho": (ncrms, nzm), "rhow": (ncrms, nz),
            "flux": (ncrms, nz) + t}


def _host_ptrs(arrs, dims, dt, writable=()):
    """Pointers of host arrays after checking dtype, order AND shape against what the C side
    will copy (a wrongly shaped array would be an out-of-bounds hipMemcpy)."""
    sh = host_shapes(*dims)
    out = []
    for name, a in arrs:
        if a is None:
            out.append(None)
            continue
        p = _host_ptr(a, name, name in writable, dt)
        # a singleton axis (the reference's j = 1) and a trailing tracer axis of 1 are accepted
        got = tuple(x for x in a.shape if x != 1)
        want = tuple(x for x in sh[name] if x != 1)
        if got != want:
            raise MpdataError(-1, f"{name}: shape {tuple(a.shape)} != expected {sh[name]}")
        out.append(p)
    return out


def advect_scalar2D_host(f, u, w, rho, rhow, flux, adz):
    """The drop-in, synchronous call on HOST arrays (numpy, Fortran order,
    reference shapes): H2D + kernel + D2H, like the reference's OpenACC
    routine with its update device/host directives (:107, :241)."""
    ncrms, nx, nz, nt = _host_dims(f)
    dt = np.float32 if isinstance(f, np.ndarray) and f.dtype == np.float32 else np.float64
    fn = lib().mpdata_advect_scalar2d if dt == np.float64 else lib().mpdata_advect_scalar2d_f32
    ptrs = _host_ptrs((("f", f), ("u", u), ("w", w), ("rho", rho), ("rhow", rhow), ("adz", adz), ("flux", flux)),
                      (ncrms, nx, nz, nt), dt, writable=("f", "flux"))
    _check(fn(ncrms, nx, nz, nt, *ptrs))


def release_host_buffers():
    """Free the streams and chunk buffers the host-array call keeps for the calling thread's next call."""
    _check(lib().mpdata_release_host_buffers())


class Plan:
    """Library-owned device state + stream (reference: `!$acc enter data`,
    `update device`, `wait`, `update host`; :105-110, :237-242).  Arrays cross
    the boundary in the reference layout; the plan keeps them in its own layout
    (`layout`: LAYOUT_WAVEMAJOR for nz <= 238 -- fp32: an even ncrms --, include/mpdata_hip.h 3)."""

    def __init__(self, ncrms, nx, nz, ntracers=1, dtype=np.float64, ngpus=None, devices=None):
        """ngpus / devices: a multi-GPU plan (include/mpdata_hip.h section 3b) -- the problem
        is cut into contiguous ncrms blocks, one per GPU; upload scatters, download gathers."""
        self._p = ctypes.c_void_p()
        self._dt = np.dtype(dtype).type
        if self._dt not in (np.float64, np.float32):
            raise MpdataError(-1, f"Plan: dtype {dtype} is neither float64 nor float32")
        self._sfx = "" if self._dt == np.float64 else "_f32"
        self.dims = (int(ncrms), int(nx), int(nz), int(ntracers))
        if ngpus is None and devices is None:
            _check(getattr(lib(), "mpdata_plan_create" + self._sfx)(ncrms, nx, nz, ntracers, ctypes.byref(self._p)))
        else:
            if self._dt != np.float64:
                raise MpdataError(-1, "multi-GPU plans are fp64")
            if devices is not None:
                arr = (ctypes.c_int * len(devices))(*devices)
                _check(lib().mpdata_plan_create_multi_devices(ncrms, nx, nz, ntracers, len(devices), arr,
                                                              ctypes.byref(self._p)))
            else:
                _check(lib().mpdata_plan_create_multi(ncrms, nx, nz, ntracers, int(ngpus), ctypes.byref(self._p)))

    @property
    def ngpus(self):
        return lib().mpdata_plan_ngpus(self._p)

    @property
    def ranks_seen(self):
        """ranks the plan's RCCL communicator reports (ncclCommCount); 0 if the plan has none"""
        return lib().mpdata_plan_ranks_seen(self._p)

    def shards(self):
        """[(device, sl0, nloc)] of the plan's GPUs."""
        out = []
        for g in range(self.ngpus):
            d, a, b = ctypes.c_int(), ctypes.c_int64(), ctypes.c_int64()
            _check(lib().mpdata_plan_shard(self._p, g, ctypes.byref(d), ctypes.byref(a), ctypes.byref(b)))
            out.append((d.value, a.value, b.value))
        return out

    def transfer_stats(self):
        """Multi-GPU plans: seconds and bytes per peer link of the last upload / download."""
        ss, gs = ctypes.c_double(), ctypes.c_double()
        sb, gb, tr = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        _check(lib().mpdata_plan_transfer_stats(self._p, ctypes.byref(ss), ctypes.byref(gs), ctypes.byref(sb),
                                                ctypes.byref(gb), ctypes.byref(tr)))
        return {"scatter_s": ss.value, "gather_s": gs.value, "scatter_bytes_per_peer": sb.value,
                "gather_bytes_per_peer": gb.value, "transport": ("rccl", "p2p", "direct")[tr.value]}

    @property
    def layout(self):
        return lib().mpdata_plan_layout(self._p)

    @property
    def device(self):
        return lib().mpdata_plan_device(self._p)

    def upload(self, f, u, w, rho, rhow, adz, flux=None):
        ptrs = _host_ptrs((("f", f), ("u", u), ("w", w), ("rho", rho), ("rhow", rhow), ("adz", adz), ("flux", flux)),
                          self.dims, self._dt)
        _check(getattr(lib(), "mpdata_plan_upload" + self._sfx)(self._p, *ptrs))

    def run(self, first_tracer=None, ntracers=None):
        if first_tracer is None:
            _check(lib().mpdata_plan_run(self._p))
        else:
            _check(lib().mpdata_plan_run_tracers(self._p, int(first_tracer), int(1 if ntracers is None else ntracers)))

    def run_uw(self, u, w, first_tracer=0, ntracers=None):
        """One step on fresh velocities: u, w = reference-layout DEVICE tensors (reversed-axes torch
        layout), f and the rest stay in the plan (mpdata_plan_run_uw)."""
        ncrms, nx, nz, nt = self.dims
        sh = shapes(ncrms, nx, nz, 1)
        pu, pw = _dev_ptr(u, sh["u"], "u", self._tdt()), _dev_ptr(w, sh["w"], "w", self._tdt())
        _check(lib().mpdata_plan_run_uw(self._p, int(first_tracer), int(nt - first_tracer if ntracers is None else ntracers), pu, pw))

    def sync(self):
        _check(lib().mpdata_plan_sync(self._p))

    def download(self, f, flux):
        ptrs = _host_ptrs((("f", f), ("flux", flux)), self.dims, self._dt, writable=("f", "flux"))
        _check(getattr(lib(), "mpdata_plan_download" + self._sfx)(self._p, *ptrs))

    def _tdt(self):
        import torch
        return torch.float64 if self._dt == np.float64 else torch.float32

    def import_device(self, f=None, u=None, w=None, rho=None, rhow=None, adz=None, flux=None, first_tracer=0):
        """Reference-layout DEVICE tensors (reversed-axes torch layout) -> the plan, on the plan's
        stream.  None = keep what the plan has.  f / flux: ([ntr,] nzm, nx+6, ncrms) /
        ([ntr,] nz, ncrms) covering tracers first_tracer .. first_tracer+ntr-1."""
        ncrms, nx, nz, nt = self.dims
        ntr = 1
        for t in (f, flux):
            if t is not None and t.dim() == (4 if t is f else 3):
                ntr = t.shape[0]
        sh = shapes(ncrms, nx, nz, ntr)
        args = []
        for name, t in (("f", f), ("u", u), ("w", w), ("rho", rho), ("rhow", rhow), ("adz", adz), ("flux", flux)):
            args.append(None if t is None else _dev_ptr(t, sh[name], name, self._tdt()))
        _check(lib().mpdata_plan_import_device(self._p, *args, int(first_tracer), int(ntr)))

    def export_device(self, f=None, flux=None, first_tracer=0):
        """The plan's f / flux of tracers first_tracer.. -> reference-layout device tensors."""
        ncrms, nx, nz, nt = self.dims
        ntr = 1
        for t in (f, flux):
            if t is not None and t.dim() == (4 if t is f else 3):
                ntr = t.shape[0]
        sh = shapes(ncrms, nx, nz, ntr)
        pf = None if f is None else _dev_ptr(f, sh["f"], "f", self._tdt())
        pl = None if flux is None else _dev_ptr(flux, sh["flux"], "flux", self._tdt())
        _check(lib().mpdata_plan_export_device(self._p, pf, pl, int(first_tracer), int(ntr)))

    def set_stream(self, stream=None):
        """Run on a torch stream (default: torch's current stream) from now on."""
        _check(lib().mpdata_plan_set_stream(self._p, _stream_handle(stream)))

    def set_timing(self, on):
        """the plan's own event pair around every run (last_kernel_ms) on / off (mpdata_plan_set_timing)"""
        _check(lib().mpdata_plan_set_timing(self._p, int(bool(on))))

    def last_kernel_ms(self):
        ms = ctypes.c_double()
        _check(lib().mpdata_plan_last_kernel_ms(self._p, ctypes.byref(ms)))
        return ms.value

    def close(self):
        if self._p:
            lib().mpdata_plan_destroy(self._p)
            self._p = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fill_synthetic(t, name, seed, dist, ncrms_global=None, sl0=0, stream=None):
    """Fill device tensor `t` (reversed-axes layout, last axis = local CRM
    instances) with the synthetic law of array `name`."""
    nloc = t.shape[-1]
    rows = t.numel() // nloc
    import torch
    ng = nloc if ncrms_global is None else ncrms_global
    fn = lib().mpdata_fill_synthetic_device if t.dtype == torch.float64 else lib().mpdata_fill_synthetic_f32_device
    if t.dtype not in (torch.float64, torch.float32) or not t.is_contiguous():
        raise MpdataError(-1, f"{name}: need a contiguous float64 or float32 device tensor")
    _check(fn(ctypes.c_void_p(t.data_ptr()), SID[name], rows, ng, sl0, nloc, seed, dist,
              _stream_handle(stream)))


def _shard_check(t, name):
    import torch
    if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.float64, torch.float32)):
        raise MpdataError(-1, f"{name}: need a contiguous float64 or float32 device tensor")


def _shard_dims(full, shard):
    """rows, ncrms, nloc IN UNITS OF 8 BYTES for the fp64 pack kernel: an fp32 tensor is moved as
    pairs of elements, which needs even ncrms, nloc and sl0."""
    _shard_check(full, "full")
    _shard_check(shard, "shard")
    if full.dtype != shard.dtype or tuple(full.shape[:-1]) != tuple(shard.shape[:-1]):
        raise MpdataError(-1, f"shard {tuple(shard.shape)} {shard.dtype} does not match full "
                              f"{tuple(full.shape)} {full.dtype}")
    return full.numel() // full.shape[-1], full.shape[-1], shard.shape[-1]


def pack_shard(full, sl0, nloc, out=None, stream=None):
    """Contiguous copy of CRM instances [sl0, sl0+nloc) of a device tensor (float64, or
    float32 with even ncrms, nloc, sl0)."""
    import torch
    if out is None:
        out = torch.empty(full.shape[:-1] + (nloc,), dtype=full.dtype, device=full.device)
    rows, ncrms, nl = _shard_dims(full, out)
    if nl != nloc:
        raise MpdataError(-1, f"out holds {nl} instances, asked for {nloc}")
    if full.dtype == torch.float32:
        if (ncrms | nloc | sl0) & 1:
            raise MpdataError(-1, "float32 shards need even ncrms, nloc and sl0")
        ncrms, nloc, sl0 = ncrms // 2, nloc // 2, sl0 // 2
    _check(lib().mpdata_pack_shard_device(ctypes.c_void_p(full.data_ptr()),
                                          ctypes.c_void_p(out.data_ptr()), rows, ncrms, sl0, nloc,
                                          _stream_handle(stream)))
    return out


def unpack_shard(full, shard, sl0, stream=None):
    import torch
    rows, ncrms, nloc = _shard_dims(full, shard)
    if full.dtype == torch.float32:
        if (ncrms | nloc | sl0) & 1:
            raise MpdataError(-1, "float32 shards need even ncrms, nloc and sl0")
        ncrms, nloc, sl0 = ncrms // 2, nloc // 2, sl0 // 2
    _check(lib().mpdata_unpack_shard_device(ctypes.c_void_p(full.data_ptr()),
                                            ctypes.c_void_p(shard.data_ptr()), rows, ncrms, sl0,
                                            nloc, _stream_handle(stream)))
