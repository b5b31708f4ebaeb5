"""CPU tests of the drop-in boundary: libmpdata_hip.so loads without a GPU,
exports every symbol include/mpdata_hip.h declares, and rejects bad arguments
before touching the device.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mpdata_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpdata_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for s in ("mpdata_advect_scalar2d", "mpdata_advect_scalar2d_device", "mpdata_plan_create",
              "mpdata_plan_upload", "mpdata_plan_run", "mpdata_plan_sync", "mpdata_plan_download",
              "mpdata_plan_destroy", "mpdata_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol(mpdata):
    L = ctypes.CDLL(mpdata.lib_path())
    for s in declared_symbols():
        assert hasattr(L, s), f"{s} declared in include/mpdata_hip.h but not exported"


def test_argument_errors_without_device(mpdata):
    L = mpdata.lib()
    one = ctypes.c_void_p(8)  # never dereferenced: validation comes first
    # nz < 3 (reference needs kc != kb, :569), ncrms < 1, nx < 1, ntracers < 1
    for (n, nx, nz, nt) in ((4, 8, 2, 1), (0, 8, 6, 1), (4, 0, 6, 1), (4, 8, 6, 0)):
        rc = L.mpdata_advect_scalar2d_device(n, nx, nz, nt, one, one, one, one, one, one, one, None)
        assert rc == -1, (n, nx, nz, nt)
        assert b"bad sizes" in L.mpdata_last_error()
    rc = L.mpdata_advect_scalar2d_device(4, 8, 6, 1, None, one, one, one, one, one, one, None)
    assert rc == -1
    # nz beyond the wave-major forms (x-marching kernels: 64, window form: 238) AND nx beyond the widest k-marching
    # tiling -> MPDATA_EUNSUPPORTED
    rc = L.mpdata_advect_scalar2d_device(4, 4000, 300, 1, one, one, one, one, one, one, one, None)
    assert rc == -2
    p = ctypes.c_void_p()
    assert L.mpdata_plan_create(4, 8, 2, 1, ctypes.byref(p)) == -1 and not p.value


def test_algorithmic_bytes_formula(mpdata):
    # B1 = 8*nzm*(4nx+23) per CRM; BT = 8*nzm*(T(2nx+11)+2nx+12)
    assert mpdata.algorithmic_bytes(1, 32, 28, 1) == 32616
    assert mpdata.algorithmic_bytes(1, 32, 28, 25) == 421416
    assert mpdata.algorithmic_bytes(65536, 32, 28, 1) == 65536 * 32616


def test_variant_switch(mpdata):
    prev = mpdata.set_variant(mpdata.VARIANT_FAST)
    assert mpdata.get_variant() == mpdata.VARIANT_FAST
    mpdata.set_variant(mpdata.VARIANT_EXACT)
    assert mpdata.get_variant() == mpdata.VARIANT_EXACT
    mpdata.set_variant(prev)


def test_no_cpu_fallback_in_product(mpdata):
    """The product library must not link or reference anything under oracle/."""
    import subprocess
    out = subprocess.run(["nm", "-D", mpdata.lib_path()], capture_output=True, text=True).stdout
    assert "mpdata_oracle" not in out
    pkg = os.path.join(ROOT, "codesign-kernels_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".F90", ".f90")):
                src = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn
                assert "libmpdata_oracle" not in src, fn


def test_default_kernels_do_not_spill():
    """The x-marching and wave-major kernels must fit their register budget (128 VGPRs, 4 waves
    per SIMD): a spill shows up as scratch traffic on top of the algorithmic HBM bytes.
    The build writes hipcc's -Rpass-analysis=kernel-resource-usage report next to the
    objects."""
    import glob
    reports = glob.glob(os.path.join(ROOT, "codesign-kernels_amd", "csrc", "mpdata_kernels_*.usage.txt"))
    reports = [r for r in reports if re.fullmatch(r"mpdata_kernels_(exact|fast)_p\d\.usage\.txt", os.path.basename(r))]
    if not reports:
        pytest.skip("no resource-usage report (library not built here)")
    seen = 0
    for rep in reports:
        text = open(rep).read()
        for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)",
                             text, flags=re.S):
            name, scratch, occ = m.group(1), int(m.group(2)), int(m.group(3))
            if "wm_ks2_kernel" in name:    # nz 65 .. 90: full-width waves + tail waves in one workgroup
                seen += 1
                assert scratch == 0 and occ >= 2, f"{name}: scratch {scratch}, occupancy {occ}"
            if "wm_odd_kernel" in name:    # tracer batches with an odd count: two-tracer waves + one one-tracer wave per tile
                seen += 1
                assert scratch == 0 and occ >= 2, f"{name}: scratch {scratch}, occupancy {occ}"
            if "xmarch" in name or "wm_kernel" in name:
                seen += 1
                # template arguments of the wave-major kernel: <R, LPS, WPB, STREAM, TPW, UWREF, UWCONV, NPK>
                t = re.search(r"wm_kernelI(d|Dv2_f)Li(\d+)ELi(\d+)ELb([01])ELi(\d)ELb([01])ELb([01])ELi(\d+)EEE", name)
                assert ("wm_kernel" in name) == (t is not None), name
                lps, tpw, uwref, npk = (int(t.group(2)), int(t.group(5)), t.group(6) == "1", int(t.group(8))) if t else (0, 1, False, 0)
                exact = "mpdata_exact" in name
                x = re.search(r"xmarch_kernelI(d|f|Dv2_f)Li(\d+)ELi(\d+)ELb([01])ELb([01])ELi(\d+)EEE", name)
                assert ("xmarch_kernel" in name) == (x is not None), name
                if x:
                    npk = int(x.group(6))
                if npk:    # EXACT with the register park (nx limited vertical fluxes per lane in registers): 2 waves per SIMD
                    assert exact and scratch == 0 and occ >= 2, f"{name}: scratch {scratch}, occupancy {occ}"
                    continue
                if lps == 128:   # nz > 64: several waves per instance, each a one-instance-per-wave form
                    lps = 64
                # (the EXACT build of the u, w-ring kernel with one instance per wave -- LPS = 64, a 16-wave workgroup
                #  that caps the registers at 128 where its other tilings take 132 -- is the parity variant of
                #  mpdata_plan_run_uw for nz 33 .. 64, not a timed one: a few spilled registers are accepted there)
                assert scratch <= (32 if (exact and uwref and lps == 64) else 0), f"{name} spills {scratch} bytes/lane"
                # (the one-instance-per-wave wave-major kernels, nz 33..64, are built for 3 waves per SIMD; the
                #  two-tracers-per-wave batch kernels for 2; the EXACT build of the kernel that reads u, w from the
                #  reference layout is the parity variant of that kernel and runs one 8-wave workgroup per CU: 2)
                want = 2 if tpw == 2 else 2 if (uwref and exact) else 3 if (lps == 64 and not uwref) else 4
                assert occ >= want, f"{name} occupancy {occ} waves/SIMD"
    assert seen >= 8 + 16   # x-march tilings + wave-major kernels (4 LPS x 2 fetch modes), both variants


def test_shipped_library_is_a_production_build(mpdata):
    """mpdata_version() names every timing-ablation / experiment macro the kernels were compiled with
    (mpdata_kernels_inst.h: build_flags); some of them produce wrong results by design, and a stray
    -DMPDWM_ABL_NOCOMPUTE in CXXFLAGS would make a green, fast and wrong bench.  The library the
    package loads must have been built with none."""
    v = mpdata.version()
    assert "exact[]" in v and "fast[]" in v, v
    assert os.path.basename(mpdata.lib_path()) == "libmpdata_hip.so" or os.environ.get("MPDATA_HIP_LIB")
