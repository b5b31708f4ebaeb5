"""The Fortran driver + ISO_C_BINDING shim (codesign-kernels_amd/fortran/):
`call advect_scalar2D(f,u,w,rho,rhow,flux)` through libmpdata_hip.so, checked
against the oracle -- the compare() step of the reference's driver
(mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:679-683), done here
because the product driver carries no CPU advection routine."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "codesign-kernels_amd", "fortran", "advect")
EXE_SP = os.path.join(ROOT, "codesign-kernels_amd", "fortran", "advect_sp")   # make hip=1 single=1


def test_driver_is_built_and_links_the_c_abi():
    """CPU check: the driver exists (built by __graft_entry__.build) and binds
    the C-ABI entry points by name."""
    if not os.path.exists(EXE):
        pytest.skip("driver not built (run __graft_entry__.build())")
    syms = subprocess.run(["nm", "-D", "--undefined-only", EXE], capture_output=True, text=True).stdout
    for s in ("mpdata_advect_scalar2d", "mpdata_plan_create", "mpdata_plan_upload", "mpdata_plan_run",
              "mpdata_plan_sync", "mpdata_plan_download", "mpdata_plan_destroy", "mpdata_last_error"):
        assert s in syms, s
    assert "mpdata_oracle" not in syms
    if os.path.exists(EXE_SP):   # the fp32 build binds the *_f32 entry points
        syms = subprocess.run(["nm", "-D", "--undefined-only", EXE_SP], capture_output=True, text=True).stdout
        for s in ("mpdata_advect_scalar2d_f32", "mpdata_plan_create_f32", "mpdata_plan_upload_f32",
                  "mpdata_plan_download_f32", "mpdata_plan_run"):
            assert s in syms, s
        assert "mpdata_oracle" not in syms


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dist,variant", [((100, 32, 28), 1, 0), ((48, 32, 58), 2, 0), ((64, 32, 28), 1, 1),
                                                ((40, 32, 72), 2, 0), ((33, 20, 100), 1, 1)])
def test_driver_matches_oracle(oracle, tmp_path, shape, dist, variant):
    assert os.path.exists(EXE), "Fortran driver not built"
    ncrms, nx, nz = shape
    dump = tmp_path / "out.bin"
    inp = oracle.make_inputs(ncrms, nx, nz, seed=100, dist=dist)   # the driver's init(): seed 100
    f_ref, flux_ref = oracle.advect(inp)
    ref = tmp_path / "ref.bin"                                     # what compare() reads
    with open(ref, "wb") as fh:
        fh.write(f_ref.tobytes(order="F"))
        fh.write(flux_ref.tobytes(order="F"))
    res = subprocess.run([EXE, str(ncrms), str(nx), str(nz), str(dist), str(variant), str(dump), str(ref)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "HIP Timing:" in res.stdout
    # the reference's compare() lines (:681-682), printed by the driver itself
    import re
    m = re.search(r"Relative L1 Error - f\s*:\s*([0-9.Ee+-]+)", res.stdout)
    m2 = re.search(r"Relative L1 Error - flux\s*:\s*([0-9.Ee+-]+)", res.stdout)
    assert m and m2, res.stdout
    assert float(m.group(1)) == (0.0 if variant == 0 else pytest.approx(0.0, abs=1e-14))
    assert float(m2.group(1)) == 0.0 if variant == 0 else float(m2.group(1)) < 1e-13   # (EXACT: flux bit-identical too)
    # the in-program self-check (no file needed): the drop-in call's kernel against the resident plan's kernel
    sc = res.stdout[res.stdout.index("Self-check"):]
    s1 = re.search(r"Relative L1 Error - f\s*:\s*([0-9.Ee+-]+)", sc)
    s2 = re.search(r"Relative L1 Error - flux\s*:\s*([0-9.Ee+-]+)", sc)
    assert s1 and s2, sc
    assert float(s1.group(1)) == (0.0 if variant == 0 else pytest.approx(0.0, abs=1e-14))
    assert float(s2.group(1)) == 0.0 if variant == 0 else float(s2.group(1)) < 1e-13
    raw = np.fromfile(dump, dtype=np.float64)
    f = raw[:f_ref.size].reshape(f_ref.shape, order="F")
    flux = raw[f_ref.size:].reshape(flux_ref.shape, order="F")
    nzm = nz - 1
    # the reference's own metric (:681-682)
    print("Relative L1 Error - f    :", oracle.rel_l1(f, f_ref))
    print("Relative L1 Error - flux :", oracle.rel_l1(flux[:, :nzm], flux_ref[:, :nzm]))
    if variant == 0:
        assert np.array_equal(f, f_ref)          # Fortran generator == oracle generator, HIP == oracle
    elif dist == 1:
        assert np.abs(f - f_ref).max() < 1e-12
    assert np.all(np.abs(flux[:, :nzm] - flux_ref[:, :nzm]) <= 1e-12 * np.maximum(1.0, np.abs(flux_ref[:, :nzm])))
    assert np.array_equal(flux[:, nzm], inp["flux"][:, nzm])   # level nz untouched


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dist,variant", [((100, 32, 28), 1, 0), ((101, 32, 28), 3, 0), ((64, 32, 28), 1, 1),
                                                ((64, 16, 72), 1, 0)])
def test_single_precision_driver_matches_fp32_oracle(oracle, tmp_path, shape, dist, variant):
    """`make hip=1 single=1`: rp = fp32 (the reference's precision switch, :12), through
    mpdata_advect_scalar2d_f32 and the fp32 plan API; even ncrms -> packed kernels, odd ->
    one instance per lane."""
    assert os.path.exists(EXE_SP), "single-precision Fortran driver not built"
    ncrms, nx, nz = shape
    dump = tmp_path / "out.bin"
    res = subprocess.run([EXE_SP, str(ncrms), str(nx), str(nz), str(dist), str(variant), str(dump)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "HIP Timing:" in res.stdout
    inp = oracle.make_inputs(ncrms, nx, nz, seed=100, dist=dist, dtype=np.float32)
    f_ref, flux_ref = oracle.advect(inp)
    raw = np.fromfile(dump, dtype=np.float32)
    f = raw[:f_ref.size].reshape(f_ref.shape, order="F")
    flux = raw[f_ref.size:].reshape(flux_ref.shape, order="F")
    nzm = nz - 1
    if variant == 0:
        assert np.array_equal(f, f_ref)
    else:
        assert np.abs(f.astype(np.float64) - f_ref).max() < 1e-5
    d = np.abs(flux[:, :nzm].astype(np.float64) - flux_ref[:, :nzm])
    assert np.all(d <= 2e-5 * np.maximum(1.0, np.abs(flux_ref[:, :nzm])))
    assert np.array_equal(flux[:, nzm], inp["flux"][:, nzm])


def test_driver_binds_the_multi_gpu_entry_points():
    if not os.path.exists(EXE):
        pytest.skip("driver not built (run __graft_entry__.build())")
    syms = subprocess.run(["nm", "-D", "--undefined-only", EXE], capture_output=True, text=True).stdout
    for s in ("mpdata_plan_create_multi", "mpdata_plan_transfer_stats"):
        assert s in syms, s


@pytest.mark.gpu
@pytest.mark.parametrize("ntr,ngpus,use_nml", [(3, 1, False), (1, 2, False), (4, 3, True)])
def test_driver_tracers_and_gpus(oracle, tmp_path, ntr, ngpus, use_nml):
    """./advect ... ntracers ngpus (and the namelist form): the tracer-batched call and the ncrms
    axis sharded over `ngpus` GPUs (all mapped to device 0 here: MPDATA_MULTI_DEVICES), through
    the Fortran shim -> mpdata_plan_create_multi -> scatter / run / gather.  Bitwise against the
    oracle in the EXACT variant."""
    assert os.path.exists(EXE), "Fortran driver not built"
    ncrms, nx, nz, dist = 90, 32, 28, 1
    dump = tmp_path / "out.bin"
    inp = oracle.make_inputs(ncrms, nx, nz, seed=100, dist=dist, ntracers=ntr)
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    env = dict(os.environ, MPDATA_MULTI_DEVICES=",".join(["0"] * ngpus))
    if use_nml:
        nml = tmp_path / "case.nml"
        nml.write_text(f"&advect_nml\n ncrms={ncrms}, nx={nx}, nz={nz}, dist={dist}, variant=0,\n"
                       f" ntracers={ntr}, ngpus={ngpus}, dumpfile='{dump}', reffile='-'\n/\n")
        cmd = [EXE, str(nml)]
    else:
        cmd = [EXE, str(ncrms), str(nx), str(nz), str(dist), "0", str(dump), "-", str(ntr), str(ngpus)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "HIP Timing:" in res.stdout
    if ngpus > 1:
        assert "scatter seconds" in res.stdout and "gather  seconds" in res.stdout
    # the in-program self-check (no file needed): the drop-in call's kernel against the resident plan's kernel
    sc = res.stdout[res.stdout.index("Self-check"):]
    s1 = re.search(r"Relative L1 Error - f\s*:\s*([0-9.Ee+-]+)", sc)
    s2 = re.search(r"Relative L1 Error - flux\s*:\s*([0-9.Ee+-]+)", sc)
    assert s1 and s2, sc
    assert float(s1.group(1)) == 0.0   # (EXACT)
    assert float(s2.group(1)) < 1e-13
    raw = np.fromfile(dump, dtype=np.float64)
    f = raw[:f_ref.size].reshape(f_ref.shape, order="F")
    flux = raw[f_ref.size:].reshape(flux_ref.shape, order="F")
    assert np.array_equal(f, f_ref)
    nzm = nz - 1
    assert np.all(np.abs(flux[:, :nzm] - flux_ref[:, :nzm]) <= 1e-13 * np.maximum(1.0, np.abs(flux_ref[:, :nzm])))
    assert np.array_equal(flux[:, nzm], inp["flux"][:, nzm])


def _checksums(out):
    import re
    f = re.search(r"checksum f\s*:\s*([0-9.Ee+-]+)", out)
    x = re.search(r"checksum flux\s*:\s*([0-9.Ee+-]+)", out)
    assert f and x, out
    return float(f.group(1)), float(x.group(1))


@pytest.mark.gpu
@pytest.mark.parametrize("ntr,ngpus,fake_rccl", [(1, 1, False), (3, 2, False), (2, 3, True)])
def test_driver_device_mode(oracle, ntr, ngpus, fake_rccl):
    """mode = device: the global arrays are generated on the root GPU (the law init() uses), scattered from
    there, advected, gathered and summed on the device -- no host array of the problem's size (what
    BASELINE.json configs[4] needs).  Checked through the printed checksums against the oracle on the same
    law.  With ngpus > 1 every rank sits on device 0: the peer-copy transport, or -- the driver run against
    the test-only link of the library with the recording RCCL stand-in (LD_PRELOAD, MPDATA_MULTI_FORCE_RCCL) --
    the RCCL branch, which then reports its ranks."""
    assert os.path.exists(EXE), "Fortran driver not built"
    ncrms, nx, nz, dist = 96, 32, 28, 1
    inp = oracle.make_inputs(ncrms, nx, nz, seed=100, dist=dist, ntracers=ntr)
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    env = dict(os.environ, MPDATA_MULTI_DEVICES=",".join(["0"] * ngpus))
    if fake_rccl:
        fake = os.path.join(ROOT, "tests", "stubs", "libmpdata_hip_fakerccl.so")
        if not os.path.exists(fake):
            pytest.skip("tests/stubs/libmpdata_hip_fakerccl.so not built")
        env.update(LD_PRELOAD=fake, MPDATA_MULTI_FORCE_RCCL="1")
    res = subprocess.run([EXE, str(ncrms), str(nx), str(nz), str(dist), "0", "-", "-", str(ntr), str(ngpus), "device"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "mode: device" in res.stdout and "HIP kernel (hipEvent) seconds" in res.stdout
    sf, sx = _checksums(res.stdout)
    nzm = nz - 1
    assert sf == pytest.approx(float(f_ref.sum()), rel=1e-12)
    assert sx == pytest.approx(float(flux_ref[:, :nzm].sum()), rel=1e-10, abs=1e-9)
    import re
    ranks = int(re.search(r"RCCL ranks seen[^:]*:\s*(\d+)", res.stdout).group(1))
    assert ranks == (ngpus if fake_rccl else 0)
    if ngpus > 1:
        assert "scatter seconds" in res.stdout and "gather  seconds" in res.stdout


@pytest.mark.gpu
def test_park_array_of_the_device_call_is_stable_over_many_processes(oracle, tmp_path):
    """Round 5 regression: the EXACT device call parks the limited vertical fluxes in HBM where they do not fit registers
    (here: the fp32 one-instance-per-lane kernel, a 16-wave workgroup).  With the park array allocated in stream order
    around every call (hipMallocAsync / hipFreeAsync on the legacy default stream) about one process in fifty delivered
    a flux whose limited part belonged to nothing for ONE workgroup (f right).  The array is kept per host thread and
    stream now: 60 fresh processes, flux bit-identical every time."""
    assert os.path.exists(EXE_SP), "single-precision Fortran driver not built"
    ncrms, nx, nz, dist = 101, 32, 28, 3
    inp = oracle.make_inputs(ncrms, nx, nz, seed=100, dist=dist, dtype=np.float32)
    f_ref, flux_ref = oracle.advect(inp)
    dump = tmp_path / "out.bin"
    for rep in range(60):
        res = subprocess.run([EXE_SP, str(ncrms), str(nx), str(nz), str(dist), "0", str(dump)], capture_output=True, text=True,
                             timeout=120)
        assert res.returncode == 0, res.stdout + res.stderr
        raw = np.fromfile(dump, dtype=np.float32)
        f = raw[:f_ref.size].reshape(f_ref.shape, order="F")
        flux = raw[f_ref.size:].reshape(flux_ref.shape, order="F")
        assert np.array_equal(f, f_ref), rep
        assert np.array_equal(flux[:, :nz - 1], flux_ref[:, :nz - 1]), rep
