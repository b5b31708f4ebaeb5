"""nz > 64 behind the plan API (round 5; the reference's nz is a compile-time parameter of the routine, reference :8 --
CRMs with more than 64 levels exist).  An instance is then wider than a wave: 1 + ceil((nz - 64) / 58) waves share
it, each on 64 levels of its own, NOTHING crosses between them -- the routine's dependency cone has radius 3 in k
(SURVEY.md 8 a14), so a wave's results are right from its 4th to its 61st level and those ranges tile the column
(csrc/mpdata_kernel_wm_body.h, kernel form LPS = 128).  Up to round 4 such plans fell back to the k-marching kernel
of round 1 (10-30 % of the headline's rate).

Bars as everywhere: EXACT f AND flux bit-identical to the oracle (flux through the register parks at nx <= 66, the park array beyond),
FAST max|d| < 1e-12 on conditioned inputs; the output contract on the whole arrays."""
import numpy as np
import pytest

from util import max_abs, to_dev, to_host

pytestmark = pytest.mark.gpu

# nz: the first size of the form (65), seams at every parity (koff of the last wave = the even number >= nz - 64),
# the largest two-wave size (122), three waves (123 ... 180), four (181 ... 238)
# (nz <= 74 / <= 90: the last window is a 16- / 32-lane share of a wave that holds those of 4 / 2 instances -- workgroups
#  of 3 / 2 instances: ncrms at every remainder; 74 | 75 and 90 | 91 are the seams between the forms)
SHAPES = [(9, 32, 72), (4, 5, 65), (3, 7, 66), (5, 12, 67), (6, 3, 100), (3, 7, 121), (2, 9, 122), (3, 4, 123),
          (5, 12, 127), (130, 32, 72), (67, 33, 90), (7, 9, 74), (5, 11, 75), (1, 4, 70), (2, 6, 80), (11, 5, 91),
          (10, 8, 89), (3, 40, 73), (131, 13, 81),
          # above 127 levels (three and four windows; the layout kernels take 8 instances per workgroup there); 238 = 64 + 3 * 58
          # is the largest: four windows = the four waves of a workgroup
          (3, 5, 128), (2, 7, 129), (4, 6, 180), (5, 9, 181), (2, 4, 238), (18, 11, 200)]


@pytest.fixture(autouse=True)
def _defaults(mpdata):
    mpdata.set_wm_flags(0)
    mpdata.set_plan_layout(mpdata.LAYOUT_WAVEMAJOR)
    yield
    mpdata.set_variant(mpdata.VARIANT_EXACT)


def _run(M, inp, ntr=1):
    ncrms, nxp6, nzm = inp["f"].shape[:3]
    p = M.Plan(ncrms, nxp6 - 6, nzm + 1, ntr)
    assert p.layout == M.LAYOUT_WAVEMAJOR      # (no fall-back to the reference-layout plan and its k-marching kernel)
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    p.close()
    return f, flux


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", ["exact", "fast"])
@pytest.mark.parametrize("dist", [1, 3])
def test_shapes_above_64_levels(mpdata, oracle, shape, variant, dist):
    M = mpdata
    var = M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST
    M.set_variant(var)
    inp = oracle.make_inputs(*shape, seed=13, dist=dist)
    f, flux = _run(M, inp)
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    nzm = shape[2] - 1
    if var == M.VARIANT_EXACT:
        assert np.array_equal(f, f_ref), f"max|df| = {max_abs(f, f_ref):.3e}"
        assert np.array_equal(flux, flux_ref), f"max|dflux| = {max_abs(flux, flux_ref):.3e}"
    elif dist == 1:
        assert max_abs(f, f_ref) < 1e-12 and max_abs(flux, flux_ref) < 1e-12
    else:
        assert oracle.rel_l1(f, f_ref) < 1e-14 and oracle.rel_l1(flux[:, :nzm], flux_ref[:, :nzm]) < 1e-14
    # output contract: columns -2 and nx+3 untouched, flux(:,nz) untouched
    assert np.array_equal(f[:, 0], inp["f"][:, 0]) and np.array_equal(f[:, -1], inp["f"][:, -1])
    assert np.array_equal(flux[:, -1], inp["flux"][:, -1])


@pytest.mark.parametrize("shape,ntr", [((9, 32, 72), 3), ((21, 7, 100), 2), ((4, 12, 125), 5), ((7, 10, 80), 4),
                                       ((5, 9, 66), 5), ((20, 6, 74), 2), ((3, 6, 150), 2), ((2, 5, 238), 3)])
@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_tracer_batches_above_64_levels(mpdata, oracle, shape, ntr, variant):
    M = mpdata
    var = M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST
    M.set_variant(var)
    base = oracle.make_inputs(*shape, seed=21, dist=3 if var == M.VARIANT_EXACT else 1)
    fs = [oracle.make_inputs(*shape, seed=210 + t, dist=3 if var == M.VARIANT_EXACT else 1)["f"] for t in range(ntr)]
    inp = dict(base)
    inp["f"] = np.asfortranarray(np.stack(fs, axis=-1))
    inp["flux"] = np.asfortranarray(np.stack([base["flux"]] * ntr, axis=-1))
    f, flux = _run(M, inp, ntr)
    for t in range(ntr):
        f_ref, flux_ref = oracle.advect(dict(base, f=fs[t].copy()), nthreads=4)
        if var == M.VARIANT_EXACT:
            assert np.array_equal(f[..., t], f_ref) and np.array_equal(flux[..., t], flux_ref)
        else:
            assert max_abs(f[..., t], f_ref) < 1e-12 and max_abs(flux[..., t], flux_ref) < 1e-12


@pytest.mark.parametrize("shape,ntr", [((5, 70, 70), 1), ((7, 67, 72), 1), ((3, 100, 127), 1), ((4, 130, 65), 1), ((5, 80, 100), 3)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_exact_with_more_columns_than_the_register_park_holds(mpdata, oracle, shape, ntr):
    """EXACT with nx > 66 at nz > 64: the register park holds 36 or 66 columns; beyond, every wave of an instance parks the
    limited fluxes of its window in HBM ([tracer][instance][wave][column][lane]) and the finishing kernel adds the rows a
    wave owns in the reference's order -- wave-major (round 5 fell back to the k-marching kernel here: 12 Gcu/s), f and
    flux bit-identical."""
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    base = oracle.make_inputs(*shape, seed=5, dist=3)
    fs = [oracle.make_inputs(*shape, seed=50 + t, dist=3)["f"] for t in range(ntr)]
    inp = dict(base)
    if ntr > 1:
        inp["f"] = np.asfortranarray(np.stack(fs, axis=-1))
        inp["flux"] = np.asfortranarray(np.stack([base["flux"]] * ntr, axis=-1))
    else:
        inp["f"] = fs[0]
    p = M.Plan(*shape, ntr)
    assert p.layout == M.LAYOUT_WAVEMAJOR
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    p.close()
    for t in range(ntr):
        f_ref, flux_ref = oracle.advect(dict(base, f=fs[t].copy()), nthreads=4)
        ft, flt = (f[..., t], flux[..., t]) if ntr > 1 else (f, flux)
        assert np.array_equal(ft, f_ref) and np.array_equal(flt, flux_ref)


def test_exact_fp32_with_more_columns_than_the_register_park_holds(mpdata, oracle):
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    shape = (6, 75, 90)
    inp = oracle.make_inputs(*shape, seed=9, dist=1, dtype=np.float32)
    p = M.Plan(*shape, 1, dtype=np.float32)
    assert p.layout == M.LAYOUT_WAVEMAJOR
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    p.close()
    f_ref, flux_ref = oracle.advect(inp)
    assert np.array_equal(f, f_ref) and np.array_equal(flux, flux_ref)


def test_run_uw_and_device_import_above_64_levels(mpdata, oracle):
    """device-side import / export and a step on fresh reference-layout velocities (conversion path) of such a plan"""
    import torch
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    shape = (34, 9, 80)
    inp = oracle.make_inputs(*shape, seed=3, dist=3)
    other = oracle.make_inputs(*shape, seed=77, dist=3)
    d = {k: to_dev(v) for k, v in inp.items()}
    ou, ow = to_dev(other["u"]), to_dev(other["w"])
    p = M.Plan(*shape, 1)
    assert p.layout == M.LAYOUT_WAVEMAJOR
    p.import_device(d["f"], ou, ow, d["rho"], d["rhow"], d["adz"], d["flux"])
    p.run_uw(d["u"], d["w"])
    p.sync()
    fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
    p.export_device(fo, flo)
    p.sync(); torch.cuda.synchronize()
    p.close()
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    assert np.array_equal(to_host(fo), f_ref) and np.array_equal(to_host(flo), flux_ref)


def test_sharded_plan_above_64_levels(mpdata, oracle, monkeypatch):
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    monkeypatch.setenv("MPDATA_MULTI_XFER", "direct")
    shape = (21, 6, 70)
    inp = oracle.make_inputs(*shape, seed=8, dist=3)
    p = M.Plan(*shape, 1, devices=[0, 0])
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    p.close()
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    assert np.array_equal(f, f_ref) and np.array_equal(flux, flux_ref)


def test_4096_instances_of_72_levels_and_its_rate(mpdata, oracle):
    """4096 x 32 x 72 (the size VERDICT r04 names): EXACT f, flux bitwise against the oracle; FAST within 1e-12; the
    FAST plan run at 65536 x 32 x 72 reaches >= 0.40 of the HBM roofline (the k-marching fall-back: 0.1-0.2)."""
    import torch
    M = mpdata
    shape = (4096, 32, 72)
    inp = oracle.make_inputs(*shape, seed=100, dist=1)
    f_ref, flux_ref = oracle.advect(inp, nthreads=16)
    M.set_variant(M.VARIANT_EXACT)
    f, flux = _run(M, inp)
    assert np.array_equal(f, f_ref) and np.array_equal(flux, flux_ref)
    M.set_variant(M.VARIANT_FAST)
    f, flux = _run(M, inp)
    assert max_abs(f, f_ref) < 1e-12 and max_abs(flux, flux_ref) < 1e-12
    # rate, resident data, cold field sets
    ncrms, nx, nz = 32768, 32, 72
    sh = M.shapes(ncrms, nx, nz, 1)
    d = {k: torch.empty(sh[k], dtype=torch.float64, device="cuda:0") for k in ("f", "u", "w", "rho", "rhow", "adz", "flux")}
    for k in d:
        M.fill_synthetic(d[k], k, 100, 1)
    plans = []
    for s_ in range(6):
        p = M.Plan(ncrms, nx, nz, 1)
        p.set_stream(); p.set_timing(False)
        M.fill_synthetic(d["f"], "f", 100 + s_, 1)
        p.import_device(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["adz"], d["flux"])
        plans.append(p)
    for i in range(12):
        plans[i % 6].run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(24):
        plans[i % 6].run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 24
    for p in plans:
        p.close()
    frac = M.algorithmic_bytes(ncrms, nx, nz, 1) / (ms * 1e-3) / 8e12
    print(f"nz = 72, ncrms = {ncrms}: {ms:.4f} ms per plan run = {frac:.3f} of 8 TB/s")
    assert frac >= 0.40, (ms, frac)


@pytest.mark.parametrize("shape", [(10, 9, 72), (4, 5, 65), (6, 12, 127), (130, 32, 72), (10, 7, 80), (14, 5, 90), (4, 5, 200), (6, 3, 130)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_fp32_plans_above_64_levels(mpdata, oracle, shape, variant):
    """fp32 plans with an even ncrms (two adjacent instances per lane) at nz > 64: the same several-waves-per-instance form
    in its `float2` instantiation; EXACT f bit-identical to the fp32 oracle.  (Up to round 4: MPDATA_EUNSUPPORTED.)"""
    M = mpdata
    F32 = np.float32
    var = M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST
    M.set_variant(var)
    inp = oracle.make_inputs(*shape, seed=21, dist=1, dtype=F32)
    f_ref, flux_ref = oracle.advect(inp)
    p = M.Plan(*shape, 1, dtype=F32)
    assert p.layout == M.LAYOUT_WAVEMAJOR
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    p.close()
    nzm = shape[2] - 1
    if var == M.VARIANT_EXACT:
        assert np.array_equal(f, f_ref) and np.array_equal(flux, flux_ref)
    else:
        assert np.abs(f.astype(np.float64) - f_ref).max() < 1e-5
        d = np.abs(flux[:, :nzm].astype(np.float64) - flux_ref[:, :nzm])
        assert np.all(d <= 2e-5 * np.maximum(1.0, np.abs(flux_ref[:, :nzm])))
    assert np.array_equal(flux[:, nzm], inp["flux"][:, nzm])


@pytest.mark.parametrize("variant", ["exact", "fast"])
@pytest.mark.parametrize("nz,ntr", [(72, 1), (127, 1), (72, 4)])
def test_sibling_waves_do_not_overtake_each_other(mpdata, oracle, variant, nz, ntr):
    """The waves of an instance read each other's output levels as halo levels and f is updated in place: a wave may
    store a column only when its siblings have fetched it (one workgroup per instance group, a barrier per column pair).
    Without it 4096 x 32 x 72 failed under load while every small shape passed.  Here: 16384 (8192) instances, 12 launches
    from the same pristine state while a second stream saturates HBM with copies -- every launch bit-identical to the
    first, the first equal to the oracle on sampled blocks."""
    import torch
    M = mpdata
    var = M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST
    M.set_variant(var)
    ncrms, nx = (16384 if nz == 72 else 8192), 32
    sh = M.shapes(ncrms, nx, nz, 1)
    d = {k: torch.empty(sh[k], dtype=torch.float64, device="cuda:0") for k in ("u", "w", "rho", "rhow", "adz", "flux")}
    for k in d:
        M.fill_synthetic(d[k], k, 100, 1)
    fs = []
    for t in range(ntr):
        f = torch.empty(sh["f"], dtype=torch.float64, device="cuda:0")
        M.fill_synthetic(f, "f", 100 + t, 1)
        fs.append(f)
    p = M.Plan(ncrms, nx, nz, ntr)
    assert p.layout == M.LAYOUT_WAVEMAJOR
    p.set_stream()
    p.import_device(None, d["u"], d["w"], d["rho"], d["rhow"], d["adz"], None)
    side = torch.cuda.Stream()
    junk_a = torch.empty(64 * 2**20, dtype=torch.float64, device="cuda:0").fill_(1.0)
    junk_b = torch.empty_like(junk_a)
    fo, flo = torch.empty_like(fs[0]), torch.empty_like(d["flux"])
    first = None
    for it in range(12):
        for t in range(ntr):
            p.import_device(fs[t], flux=d["flux"], first_tracer=t)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(6):
                junk_b.copy_(junk_a)
        p.run()
        torch.cuda.synchronize()
        outs = []
        for t in range(ntr):
            p.export_device(fo, flo, first_tracer=t)
            torch.cuda.synchronize()
            outs.append((fo.clone(), flo.clone()))
        if first is None:
            first = outs
        else:
            for t in range(ntr):
                assert torch.equal(outs[t][0], first[t][0]) and torch.equal(outs[t][1], first[t][1]), f"launch {it}, tracer {t}"
    p.close()
    nzm = nz - 1
    for a0 in (0, ncrms // 2 - 7, ncrms - 33):
        blk = {k: to_host(v[..., a0:a0 + 33]) for k, v in d.items()}
        for t in range(ntr):
            one = dict(blk, f=to_host(fs[t][..., a0:a0 + 33]))
            fr, flr = oracle.advect(one, nthreads=4)
            fg, flg = to_host(first[t][0][..., a0:a0 + 33]), to_host(first[t][1][..., a0:a0 + 33])
            if var == M.VARIANT_EXACT:
                assert np.array_equal(fg, fr) and np.array_equal(flg, flr)
            else:
                assert max_abs(fg, fr) < 1e-12 and max_abs(flg[:, :nzm], flr[:, :nzm]) < 1e-12


# ---- calls on reference-layout DEVICE arrays (mpdata_advect_scalar2d_device, the Fortran drivers' call) at nz 65 .. 238:
#      through a wave-major plan the library keeps per host thread (csrc/mpdata_plan.hip: staged_device_call) -- up to
#      round 5 the k-marching kernel (13-16 Gcu/s; fp32 and nx > 140: MPDATA_EUNSUPPORTED)
def _stack(oracle, shape, ntr, dtype=np.float64, dist=3):
    base = oracle.make_inputs(*shape, seed=5, dist=dist, dtype=dtype)
    fs = [oracle.make_inputs(*shape, seed=50 + t, dist=dist, dtype=dtype)["f"] for t in range(ntr)]
    inp = dict(base)
    if ntr > 1:
        inp["f"] = np.asfortranarray(np.stack(fs, axis=-1))
        inp["flux"] = np.asfortranarray(np.stack([base["flux"]] * ntr, axis=-1))
    else:
        inp["f"] = fs[0]
    return base, fs, inp


@pytest.mark.parametrize("shape,ntr", [((9, 32, 72), 1), ((4, 150, 70), 1), ((3, 7, 127), 1), ((21, 12, 100), 3), ((5, 70, 66), 2),
                                       ((3, 7, 200), 1), ((2, 70, 238), 2)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_device_call_above_64_levels(mpdata, oracle, shape, ntr, variant):
    """whole arrays against the oracle (f's halo columns and flux(:, nz) are the caller's): EXACT bit-identical incl. flux,
    FAST < 1e-12; nx = 150 is beyond the widest k-marching tiling (an error up to round 5)"""
    from util import run_hip
    M = mpdata
    M.set_variant(M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST)
    base, fs, inp = _stack(oracle, shape, ntr, dist=3 if variant == "exact" else 1)
    f, flux = run_hip(M, inp)
    for t in range(ntr):
        f_ref, flux_ref = oracle.advect(dict(base, f=fs[t].copy()), nthreads=4)
        ft, flt = (f[..., t], flux[..., t]) if ntr > 1 else (f, flux)
        if variant == "exact":
            assert np.array_equal(ft, f_ref) and np.array_equal(flt, flux_ref)
        else:
            assert max_abs(ft, f_ref) < 1e-12 and max_abs(flt, flux_ref) < 1e-12
    M.release_host_buffers()


def test_fp32_device_call_above_64_levels(mpdata, oracle):
    from util import run_hip
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    for shape in ((6, 10, 65), (10, 9, 72), (8, 40, 127)):
        inp = oracle.make_inputs(*shape, seed=3, dist=1, dtype=np.float32)
        f, flux = run_hip(M, inp)
        f_ref, flux_ref = oracle.advect(inp)
        assert np.array_equal(f, f_ref) and np.array_equal(flux, flux_ref), shape
    M.release_host_buffers()


def test_device_calls_of_changing_shape_stream_and_variant(mpdata, oracle):
    """the plan behind these calls is kept per host thread and rebuilt when the shape, the tracer count, the precision or
    the variant changes; a call on another stream waits for the previous one's work on the plan's arrays; releasing the
    buffers in between is harmless; the same call twice gives the same answer (the plan's state does not leak)"""
    import torch
    M = mpdata
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cases = [((12, 9, 72), 1, "exact", None), ((12, 9, 72), 1, "exact", s1), ((12, 9, 72), 2, "exact", s2),
             ((7, 20, 100), 1, "fast", s1), ((12, 9, 72), 1, "fast", None), ((12, 9, 72), 1, "exact", s2)]
    for n, (shape, ntr, variant, st) in enumerate(cases):
        M.set_variant(M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST)
        base, fs, inp = _stack(oracle, shape, ntr, dist=1)
        d = {k: to_dev(v) for k, v in inp.items()}
        torch.cuda.synchronize()
        M.advect_scalar2D(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"], stream=st)
        torch.cuda.synchronize()
        f, flux = to_host(d["f"]), to_host(d["flux"])
        for t in range(ntr):
            f_ref, flux_ref = oracle.advect(dict(base, f=fs[t].copy()), nthreads=4)
            ft, flt = (f[..., t], flux[..., t]) if ntr > 1 else (f, flux)
            if variant == "exact":
                assert np.array_equal(ft, f_ref) and np.array_equal(flt, flux_ref), n
            else:
                assert max_abs(ft, f_ref) < 1e-12 and max_abs(flt, flux_ref) < 1e-12, n
        if n == 2:
            M.release_host_buffers()


def test_direct_device_call_above_64_levels_still_there(oracle):
    """MPDATA_DEVICE_CALL=direct (read once per process: a child): the k-marching kernel as up to round 5, same results"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import json, numpy as np, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import codesign_kernels_amd as M\n"
        "from oracle import oracle as O\n"
        "from util import run_hip\n"
        "O.build_lib(); M.set_variant(M.VARIANT_EXACT)\n"
        "inp = O.make_inputs(9, 32, 72, seed=5, dist=3)\n"
        "f, fl = run_hip(M, inp)\n"
        "fr, flr = O.advect(inp, nthreads=2)\n"
        "print('RESULT ' + json.dumps({'ok': bool(np.array_equal(f, fr)) and bool(np.array_equal(fl, flr))}))\n"
    ) % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MPDATA_DEVICE_CALL="direct"), capture_output=True,
                       text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])["ok"]


def test_arrays_larger_than_4GiB_at_72_levels(mpdata, oracle):
    """ncrms = 210000 at 32 x 72: f, u, w 4.4-4.5 GB each -- the plan (descriptors per wave: tiles of the tail waves are up to
    three tile strides apart, still far below 4 GiB), the layout kernels' import / export at nz > 64, and the device call
    through the calling thread's plan: sampled blocks against the oracle, the output contract on the whole arrays."""
    import torch
    from test_plan_wavemajor import _full_size_check
    M = mpdata
    ncrms, nx, nz = 210000, 32, 72
    blocks = ((0, 40), (99999, 37), (ncrms - 21, 21))
    _full_size_check(M, oracle, M.VARIANT_EXACT, ncrms, nx, nz, 1, blocks)
    # the same problem as ONE call on reference-layout device arrays
    M.set_variant(M.VARIANT_EXACT)
    sh = M.shapes(ncrms, nx, nz, 1)
    d = {k: torch.empty(s, dtype=torch.float64, device="cuda:0") for k, s in sh.items()}
    for k in ("f", "u", "w", "rho", "rhow", "adz", "flux"):
        M.fill_synthetic(d[k], k, 100, oracle.DIST_CONDITIONED)
    top = d["flux"][nz - 1].clone()
    M.advect_scalar2D(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"])
    torch.cuda.synchronize()
    assert torch.equal(d["flux"][nz - 1], top) and bool(torch.isfinite(d["f"]).all())
    for s0, n in blocks:
        inp = oracle.make_inputs(n, nx, nz, seed=100, dist=oracle.DIST_CONDITIONED, ncrms_global=ncrms, sl0=s0)
        f_ref, flux_ref = oracle.advect(inp)
        assert np.array_equal(to_host(d["f"][..., s0:s0 + n]), f_ref), s0
        assert np.array_equal(to_host(d["flux"][..., s0:s0 + n]), flux_ref), s0
    del d
    M.release_host_buffers()
    torch.cuda.empty_cache()


def test_above_238_levels_the_reference_layout(mpdata, oracle):
    """nz > 238 (more than four windows): the plan keeps the reference layout and runs the k-marching kernel -- correct,
    a tenth of the rate"""
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    shape = (5, 6, 239)
    inp = oracle.make_inputs(*shape, seed=5, dist=3)
    p = M.Plan(*shape, 1)
    assert p.layout == M.LAYOUT_REFERENCE
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    p.close()
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    assert np.array_equal(f, f_ref) and np.array_equal(flux, flux_ref)


@pytest.mark.parametrize("shape,ntr", [((10, 9, 72), 3), ((8, 7, 80), 2), ((6, 5, 130), 4), ((12, 40, 70), 2)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_fp32_tracer_batches_above_64_levels(mpdata, oracle, shape, ntr, variant):
    """fp32 tracer batches at nz > 64 (FAST: two tracers per wave in the `float2` forms of the window and tail kernels;
    EXACT: one tracer per wave, register park at nx <= 36, 66-column park / park array beyond): every tracer against a
    single-tracer call of the fp32 oracle"""
    M = mpdata
    F32 = np.float32
    var = M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST
    M.set_variant(var)
    base = oracle.make_inputs(*shape, seed=21, dist=1, dtype=F32)
    fs = [oracle.make_inputs(*shape, seed=300 + t, dist=1, dtype=F32)["f"] for t in range(ntr)]
    inp = dict(base, f=np.asfortranarray(np.stack(fs, axis=-1)), flux=np.asfortranarray(np.stack([base["flux"]] * ntr, axis=-1)))
    p = M.Plan(*shape, ntr, dtype=F32)
    assert p.layout == M.LAYOUT_WAVEMAJOR
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    p.close()
    nzm = shape[2] - 1
    for t in range(ntr):
        f_ref, flux_ref = oracle.advect(dict(base, f=fs[t].copy()))
        if var == M.VARIANT_EXACT:
            assert np.array_equal(f[..., t], f_ref) and np.array_equal(flux[..., t], flux_ref), t
        else:
            assert np.abs(f[..., t].astype(np.float64) - f_ref).max() < 1e-5, t
            d = np.abs(flux[:, :nzm, t].astype(np.float64) - flux_ref[:, :nzm])
            assert np.all(d <= 2e-5 * np.maximum(1.0, np.abs(flux_ref[:, :nzm]))), t
