"""GPU parity tests of mpdata_plan_run_uw (include/mpdata_hip.h section 3): one step on FRESH
reference-layout device u, w while f stays in the plan.  One fp64 tracer of a wave-major plan goes
through the kernel that reads u, w straight from the reference layout (template option UWREF of
mpdata_kernel_wm_body.h: 16-byte LDS-DMA of 128-byte row segments into a ring shared by the
workgroup; nz 33 .. 64: one instance per wave, a 16-wave workgroup); tracer batches send their first tracer
through the converting form of that kernel (nz <= 32); other cases (odd ncrms, unaligned bases, fp32) convert u, w first.

The plans are uploaded with OTHER velocities than the ones the step is run on, so a kernel that
read the plan's own u, w would fail.  Bars as in test_plan_wavemajor.py: EXACT -> f bit-identical
to the oracle (mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:477-642 restated), flux
within 1e-13; FAST -> max|d| < 1e-12 on conditioned inputs / rel-L1 < 1e-14 on raw ones.
"""
import numpy as np
import pytest

from util import max_abs, to_dev, to_host

pytestmark = pytest.mark.gpu

TOL_ABS = 1e-12
TOL_RELL1 = 1e-14
FLUX_RTOL = 0.0   # (EXACT plans: bit-identical since round 4)


def flux_close(flux, flux_ref):
    """EXACT through the plan API (round 4): flux(:, 1:nzm) is BIT-IDENTICAL to the reference -- the kernels park the limited
    vertical fluxes and a finishing kernel adds them onto the upwind sum one by one in the reference's order (:545, :624) --
    and flux(:, nz) is untouched."""
    return bool(np.array_equal(flux, flux_ref))


@pytest.fixture(scope="module")
def M(mpdata):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    mpdata.set_wm_flags(0)
    yield mpdata
    mpdata.set_variant(mpdata.VARIANT_EXACT)


def run_uw_case(M, oracle, shape, variant, dist, seed, ntr=1, misalign=False):
    """upload (f, OTHER u/w, rho, rhow, adz) -> run_uw(u, w device tensors) -> download; compares
    every tracer with the oracle on (f, u, w)."""
    import torch
    ncrms, nx, nz = shape
    var = M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST
    M.set_variant(var)
    inp = oracle.make_inputs(ncrms, nx, nz, seed=seed, dist=dist)
    other = oracle.make_inputs(ncrms, nx, nz, seed=seed + 977, dist=dist)
    fs = [oracle.make_inputs(ncrms, nx, nz, seed=seed + 10 + t, dist=dist)["f"] for t in range(ntr)]
    up = dict(inp)
    up["u"], up["w"] = other["u"], other["w"]
    if ntr > 1:
        up["f"] = np.asfortranarray(np.stack(fs, axis=-1))
        up["flux"] = np.asfortranarray(np.stack([inp["flux"]] * ntr, axis=-1))
    else:
        up["f"] = fs[0]
    p = M.Plan(ncrms, nx, nz, ntr)
    assert p.layout == M.LAYOUT_WAVEMAJOR
    p.upload(up["f"], up["u"], up["w"], up["rho"], up["rhow"], up["adz"], up["flux"])
    if misalign:   # bases at 8 modulo 16: the 16-byte row fetches are not possible, the library converts
        ub = torch.empty(inp["u"].size + 1, dtype=torch.float64, device="cuda:0")
        wb = torch.empty(inp["w"].size + 1, dtype=torch.float64, device="cuda:0")
        du = ub[1:].view(to_dev(inp["u"]).shape); du.copy_(to_dev(inp["u"]))
        dw = wb[1:].view(to_dev(inp["w"]).shape); dw.copy_(to_dev(inp["w"]))
        assert du.data_ptr() % 16 == 8
    else:
        du, dw = to_dev(inp["u"]), to_dev(inp["w"])
    p.run_uw(du, dw)
    p.sync()
    f = np.empty_like(up["f"], order="F")
    flux = np.empty_like(up["flux"], order="F")
    p.download(f, flux)
    p.close()
    for t in range(ntr):
        f_ref, flux_ref = oracle.advect(dict(inp, f=fs[t].copy()), nthreads=4)
        ft = f[..., t] if ntr > 1 else f
        flt = np.asfortranarray(flux[..., t]) if ntr > 1 else flux
        if var == M.VARIANT_EXACT:
            assert np.array_equal(ft, f_ref), (shape, t, max_abs(ft, f_ref))
            assert flux_close(flt, flux_ref), (shape, t, max_abs(flt, flux_ref))
        elif dist == oracle.DIST_CONDITIONED:
            assert max_abs(ft, f_ref) < TOL_ABS and max_abs(flt, flux_ref) < TOL_ABS, (shape, t)
        else:
            assert oracle.rel_l1(ft, f_ref) < TOL_RELL1, (shape, t)
            assert oracle.rel_l1(flt[:, :-1], flux_ref[:, :-1]) < TOL_RELL1, (shape, t)


# even ncrms: the UWREF kernel (ragged last workgroup: 258, 22, 2; every lane mapping 8 / 16 / 32 / 64 -- nz 33 .. 64
# is one instance per wave, a 16-wave workgroup: the reference's shipped size 48 x 32 x 58 among them; nx from 1
# upwards incl. every nx mod 6 class); odd ncrms takes the conversion path
SHAPES = [(64, 32, 28), (258, 31, 28), (16, 1, 3), (22, 5, 7), (130, 31, 12), (2, 2, 4), (48, 33, 17), (34, 34, 32),
          (18, 35, 9), (66, 36, 28), (20, 3, 32), (37, 32, 28), (10, 6, 64), (7, 2, 4), (48, 32, 58), (34, 7, 33),
          (130, 5, 45), (18, 34, 64), (2, 1, 40), (50, 33, 58)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_run_uw_single_tracer(M, oracle, shape, variant):
    run_uw_case(M, oracle, shape, variant, dist=3 if variant == "exact" else 1, seed=71)


@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_run_uw_reference_raw_inputs(M, oracle, variant):
    run_uw_case(M, oracle, (96, 32, 28), variant, dist=2, seed=5)


BATCHES = [((64, 32, 28), 3), ((258, 31, 28), 4), ((22, 5, 7), 2), ((130, 31, 12), 5), ((34, 34, 32), 3), ((2, 1, 3), 2),
           ((66, 36, 28), 25), ((50, 7, 12), 2), ((48, 32, 58), 3), ((37, 9, 12), 3)]


@pytest.mark.parametrize("shape,ntr", BATCHES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else f"T{v}")
@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_run_uw_tracer_batches(M, oracle, shape, ntr, variant):
    """ntracers > 1 on fresh velocities.  nz <= 32, even ncrms, aligned bases: the FIRST tracer of the batch goes
    through the kernel that reads u, w from the reference layout in its converting form (it writes the values it
    reads into the plan's arrays, plan layout), the others through the batch kernel behind it -- no conversion pass.
    nz 33 .. 64 (48 x 32 x 58) and odd ncrms (37): one fused u + w conversion, then the batch kernel.  The plan was
    uploaded with OTHER velocities.  Every tracer against the oracle."""
    run_uw_case(M, oracle, shape, variant, dist=3 if variant == "exact" else 1, seed=9, ntr=ntr)


def test_run_uw_tracer_batch_unaligned_bases_convert(M, oracle):
    run_uw_case(M, oracle, (64, 32, 28), "exact", dist=3, seed=9, ntr=3, misalign=True)


def test_run_uw_tracer_batch_full_size_sampled(M, oracle):
    """ncrms = 65536 x 5 tracers on fresh u, w (the converting kernel + the batch kernel at full size): sampled
    instance blocks of every tracer against the oracle, EXACT."""
    import torch
    ncrms, nx, nz, T = 65536, 32, 28, 5
    M.set_variant(M.VARIANT_EXACT)
    sh = M.shapes(ncrms, nx, nz, 1)
    d = {k: torch.empty(s, dtype=torch.float64, device="cuda:0") for k, s in sh.items()}
    for k in d:
        M.fill_synthetic(d[k], k, 100, oracle.DIST_CONDITIONED)
    p = M.Plan(ncrms, nx, nz, T)
    p.import_device(None, torch.full_like(d["u"], 0.25), torch.full_like(d["w"], -0.125), d["rho"], d["rhow"], d["adz"], None)
    ft = torch.empty_like(d["f"])
    for t in range(T):
        M.fill_synthetic(ft, "f", 100 + t, oracle.DIST_CONDITIONED)
        p.import_device(ft, flux=d["flux"], first_tracer=t)
    p.run_uw(d["u"], d["w"])
    fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
    for t in range(T):
        p.export_device(fo, flo, first_tracer=t)
        p.sync()
        for s0, n in ((0, 40), (31000 + 1, 34), (65536 - 33, 33)):
            inp = oracle.make_inputs(n, nx, nz, seed=100, dist=oracle.DIST_CONDITIONED, ncrms_global=ncrms, sl0=s0)
            inp["f"] = oracle.fill_array("f", (n, nx + 6, nz - 1), 100 + t, oracle.DIST_CONDITIONED, ncrms_global=ncrms, sl0=s0)
            f_ref, flux_ref = oracle.advect(inp)
            assert np.array_equal(to_host(fo[..., s0:s0 + n]), f_ref), (t, s0)
            assert flux_close(to_host(flo[..., s0:s0 + n]), flux_ref), (t, s0)
    # the converted velocities the batch kernel used are the caller's: a later import + plain run agrees bitwise
    p.import_device(None, d["u"], d["w"])
    M.fill_synthetic(ft, "f", 100, oracle.DIST_CONDITIONED)
    p.import_device(ft, first_tracer=3)
    p.run(3, 1)
    f3 = torch.empty_like(d["f"])
    p.export_device(f3, first_tracer=3)
    p.export_device(fo, first_tracer=0)
    p.sync()
    assert torch.equal(f3, fo)
    p.close()


def test_run_uw_one_tracer_of_a_multi_tracer_plan(M, oracle):
    """run_uw(first_tracer=1, ntracers=1) on a 3-tracer plan: the kernel that reads u, w from the reference
    layout works on that tracer alone; the other two keep their fields.  Tracer 2 is run on the plan's own
    (uploaded) velocities BEFORE the run_uw step."""
    import torch
    M.set_variant(M.VARIANT_EXACT)
    ncrms, nx, nz = 96, 32, 28
    inp = oracle.make_inputs(ncrms, nx, nz, seed=41, dist=3)
    other = oracle.make_inputs(ncrms, nx, nz, seed=42, dist=3)
    fs = [oracle.make_inputs(ncrms, nx, nz, seed=430 + t, dist=3)["f"] for t in range(3)]
    up = dict(inp, u=other["u"], w=other["w"])
    up["f"] = np.asfortranarray(np.stack(fs, axis=-1))
    up["flux"] = np.asfortranarray(np.stack([inp["flux"]] * 3, axis=-1))
    p = M.Plan(ncrms, nx, nz, 3)
    p.upload(up["f"], up["u"], up["w"], up["rho"], up["rhow"], up["adz"], up["flux"])
    p.run(2, 1)            # tracer 2 with the plan's own (other) velocities
    p.run_uw(to_dev(inp["u"]), to_dev(inp["w"]), first_tracer=1, ntracers=1)
    p.sync()
    f = np.empty_like(up["f"], order="F"); flux = np.empty_like(up["flux"], order="F")
    p.download(f, flux)
    p.close()
    assert np.array_equal(f[..., 0], fs[0])                                        # untouched
    f1_ref, _ = oracle.advect(dict(inp, f=fs[1].copy()), nthreads=4)               # fresh u, w
    assert np.array_equal(f[..., 1], f1_ref)
    f2_ref, _ = oracle.advect(dict(inp, u=other["u"], w=other["w"], f=fs[2].copy()), nthreads=4)   # the plan's u, w
    assert np.array_equal(f[..., 2], f2_ref)


@pytest.mark.parametrize("case", ["direct", "odd-ncrms", "tracer-batch", "unaligned", "reference-layout-plan", "fp32"])
def test_run_uw_leaves_no_velocities_on_any_path(M, oracle, case):
    """The post-condition of mpdata_plan_run_uw does not depend on the path the call took (the kernel that
    reads the caller's arrays leaves the plan's u, w alone, the converting paths overwrite them): afterwards
    the plan holds NO velocities -- mpdata_plan_run returns MPDATA_ESTATE until u and w are imported again,
    and then advects with the imported ones."""
    import torch
    M.set_variant(M.VARIANT_EXACT)
    ncrms, nx, nz, ntr = {"odd-ncrms": (37, 9, 12, 1), "tracer-batch": (64, 9, 12, 2)}.get(case, (64, 9, 12, 1))
    f32 = case == "fp32"
    dt = np.float32 if f32 else np.float64
    inp = oracle.make_inputs(ncrms, nx, nz, seed=61, dist=1, dtype=dt)
    other = oracle.make_inputs(ncrms, nx, nz, seed=62, dist=1, dtype=dt)
    prev = M.set_plan_layout(M.LAYOUT_REFERENCE) if case == "reference-layout-plan" else None
    try:
        p = M.Plan(ncrms, nx, nz, ntr, dtype=dt)
    finally:
        if prev is not None:
            M.set_plan_layout(prev)
    d = {k: to_dev(v) for k, v in inp.items()}
    do = {k: to_dev(v) for k, v in other.items()}
    for t in range(ntr):
        p.import_device(d["f"], do["u"], do["w"], d["rho"], d["rhow"], d["adz"], d["flux"], first_tracer=t)
    p.run(0, 1)                                   # the plan has velocities: fine
    if case == "unaligned":
        ub = torch.empty(inp["u"].size + 1, dtype=torch.float64, device="cuda:0")
        wb = torch.empty(inp["w"].size + 1, dtype=torch.float64, device="cuda:0")
        du = ub[1:].view(d["u"].shape); du.copy_(d["u"])
        dw = wb[1:].view(d["w"].shape); dw.copy_(d["w"])
    else:
        du, dw = d["u"], d["w"]
    p.run_uw(du, dw)
    with pytest.raises(M.MpdataError) as ei:
        p.run()
    assert ei.value.code == M.ESTATE and "velocities" in str(ei.value)
    with pytest.raises(M.MpdataError):
        p.run(0, 1)
    p.import_device(None, do["u"], None)          # u alone is not enough
    with pytest.raises(M.MpdataError):
        p.run()
    p.run_uw(du, dw)                              # (another run_uw needs nothing)
    p.import_device(d["f"], do["u"], do["w"])     # both again (and a fresh f for tracer 0)
    p.run(0, 1)
    fo = torch.empty_like(d["f"])
    p.export_device(fo)
    p.sync()
    p.close()
    f_ref, _ = oracle.advect(dict(inp, u=other["u"], w=other["w"]), nthreads=4)
    assert np.array_equal(to_host(fo), f_ref)


def test_run_uw_unaligned_bases_convert(M, oracle):
    run_uw_case(M, oracle, (64, 32, 28), "exact", dist=3, seed=11, misalign=True)


def test_run_uw_random_shapes(M, oracle):
    """30 seeded random shapes (even ncrms 2..400, nx 1..50, nz 3..64): EXACT f bit-identical."""
    rng = np.random.default_rng(20261104)
    for it in range(30):
        ncrms = 2 * int(rng.integers(1, 201))
        nx = int(rng.integers(1, 51))
        nz = int(rng.choice([rng.integers(3, 9), rng.integers(9, 17), rng.integers(17, 33), rng.integers(33, 65)]))
        run_uw_case(M, oracle, (ncrms, nx, nz), "exact", dist=3 if it % 3 else 1, seed=2000 + it)


def test_run_uw_agrees_with_import_then_run(M, oracle):
    """run_uw(u, w) == import_device(u, w) + run() on the same plan state: f bitwise (EXACT)."""
    import torch
    M.set_variant(M.VARIANT_EXACT)
    ncrms, nx, nz = 160, 32, 28
    inp = oracle.make_inputs(ncrms, nx, nz, seed=31, dist=3)
    d = {k: to_dev(v) for k, v in inp.items()}
    outs = []
    for direct in (True, False):
        p = M.Plan(ncrms, nx, nz, 1)
        p.import_device(d["f"], torch.zeros_like(d["u"]), torch.zeros_like(d["w"]), d["rho"], d["rhow"], d["adz"], d["flux"])
        if direct:
            p.run_uw(d["u"], d["w"])
        else:
            p.import_device(None, d["u"], d["w"])
            p.run()
        fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
        p.export_device(fo, flo)
        p.sync()
        outs.append((to_host(fo), to_host(flo)))
        p.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert flux_close(outs[0][1], outs[1][1])


@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_run_uw_full_size_sampled_against_oracle(M, oracle, variant):
    """BASELINE.json configs[2] (ncrms=65536 nx=32 nz=28): device-generated inputs, the plan is filled
    with OTHER velocities, one run_uw step, sampled instance blocks against the oracle + the output
    contract on the whole arrays."""
    import torch
    ncrms, nx, nz = 65536, 32, 28
    M.set_variant(variant)
    sh = M.shapes(ncrms, nx, nz, 1)
    d = {k: torch.empty(s, dtype=torch.float64, device="cuda:0") for k, s in sh.items()}
    for k in d:
        M.fill_synthetic(d[k], k, 100, oracle.DIST_CONDITIONED)
    p = M.Plan(ncrms, nx, nz, 1)
    p.import_device(d["f"], torch.full_like(d["u"], 0.25), torch.full_like(d["w"], -0.125), d["rho"], d["rhow"], d["adz"], d["flux"])
    p.run_uw(d["u"], d["w"])
    fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
    p.export_device(fo, flo)
    p.sync()
    assert torch.equal(fo[:, 0, :], d["f"][:, 0, :]) and torch.equal(fo[:, nx + 5, :], d["f"][:, nx + 5, :])
    assert torch.equal(flo[nz - 1], d["flux"][nz - 1])
    assert float(fo[:, 3:3 + nx, :].min()) >= 0.0
    assert bool(torch.isfinite(fo).all()) and bool(torch.isfinite(flo).all())
    for s0, n in ((0, 48), (12345 + 1, 40), (40000 - 8, 32), (65536 - 35, 35)):
        inp = oracle.make_inputs(n, nx, nz, seed=100, dist=oracle.DIST_CONDITIONED, ncrms_global=ncrms, sl0=s0)
        f_ref, flux_ref = oracle.advect(inp)
        f, flux = to_host(fo[..., s0:s0 + n]), to_host(flo[..., s0:s0 + n])
        if variant == M.VARIANT_EXACT:
            assert np.array_equal(f, f_ref) and flux_close(flux, flux_ref), s0
        else:
            assert max_abs(f, f_ref) < TOL_ABS and max_abs(flux, flux_ref) < TOL_ABS, s0
    p.close()
