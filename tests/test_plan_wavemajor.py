"""GPU parity tests of the plan API in its wave-major device layout (the kernel of
mpdata_kernel_wm_body.h + the layout conversions of mpdata_layout.hip), through the C-ABI:
plan_create / upload / run / download with HOST arrays in the reference layout, and
import_device / run_tracers / export_device with DEVICE arrays.

Bars: EXACT -> f AND flux bit-identical to the oracle and to the reference's golden outputs (round 4: the
limited vertical fluxes are parked and added in the reference's order); FAST -> max|d| < 1e-12 on
conditioned inputs, rel-L1 < 1e-14 on reference-raw inputs.
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from util import golden_cases, load_golden, max_abs, to_dev, to_host

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
    yield mpdata
    mpdata.set_variant(mpdata.VARIANT_EXACT)
    mpdata.set_plan_layout(mpdata.LAYOUT_WAVEMAJOR)


@pytest.fixture(autouse=True)
def _default_launch_switches(mpdata):
    """every test starts from, and leaves behind, the default wave-major launch (no test switch set)"""
    mpdata.set_wm_flags(0)
    yield
    mpdata.set_wm_flags(0)


def run_plan_host(M, inp, ntr=1):
    """upload -> run -> download on host arrays; returns (f, flux)."""
    ncrms, nxp6, nzm = inp["f"].shape[:3]
    p = M.Plan(ncrms, nxp6 - 6, nzm + 1, ntr)
    assert p.layout == M.LAYOUT_WAVEMAJOR
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run()
    p.sync()
    f = np.empty_like(inp["f"], order="F")
    flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    p.close()
    return f, flux


def assert_parity(M, oracle, variant, dist, f, flux, f_ref, flux_ref):
    if variant == M.VARIANT_EXACT:
        assert np.array_equal(f, f_ref), f"f differs: max|d|={max_abs(f, f_ref):.3e}"
        assert flux_close(flux, flux_ref), f"flux differs: max|d|={max_abs(flux, flux_ref):.3e}"
    elif dist == oracle.DIST_CONDITIONED:
        assert max_abs(f, f_ref) < TOL_ABS
        assert max_abs(flux, flux_ref) < TOL_ABS
    else:
        assert oracle.rel_l1(f, f_ref) < TOL_RELL1
        nzm = flux.shape[1] - 1
        assert oracle.rel_l1(flux[:, :nzm], flux_ref[:, :nzm]) < TOL_RELL1
        assert np.array_equal(flux[:, nzm], flux_ref[:, nzm])


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_wavemajor_plan_reproduces_reference_golden_bitwise(M, oracle, case):
    M.set_variant(M.VARIANT_EXACT)
    inp = oracle.make_inputs(case["ncrms"], case["nx"], case["nz"], seed=case["seed"], dist=case["dist"])
    f_ref, flux_ref = load_golden(case)
    f, flux = run_plan_host(M, inp)
    assert np.array_equal(f, f_ref), f"max|df|={max_abs(f, f_ref):.3e}"
    assert flux_close(flux, flux_ref), f"max|dflux|={max_abs(flux, flux_ref):.3e}"


# (ncrms, nx, nz): every LPS (8/16/32/64 lanes per instance), ragged ncrms (partial last tile,
# fewer tiles than a workgroup holds), odd nx (a last pair with one column), the smallest
# shapes the routine is defined for, nz at the top of each LPS class, the shipped size
SHAPES = [(64, 32, 28), (48, 32, 58), (1, 1, 3), (3, 1, 3), (7, 2, 4), (5, 3, 8), (19, 5, 9), (33, 7, 16),
          (37, 32, 17), (129, 9, 32), (21, 33, 33), (10, 6, 64), (131, 4, 28), (258, 31, 28), (100, 66, 28), (40, 70, 28),
          (33, 37, 58)]   # (nx: <= 36 / <= 66 the two register-park instantiations of EXACT, beyond: the park array)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", ["exact", "fast"])
@pytest.mark.parametrize("dist", [1, 3])
def test_wavemajor_plan_shapes(M, oracle, shape, variant, dist):
    var = M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST
    M.set_variant(var)
    inp = oracle.make_inputs(*shape, seed=11, dist=dist)
    f, flux = run_plan_host(M, inp)
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    assert_parity(M, oracle, var, dist, f, flux, f_ref, flux_ref)
    # output contract: columns -2 and nx+3 untouched, flux(:,nz) untouched
    assert np.array_equal(f[:, 0], inp["f"][:, 0]) and np.array_equal(f[:, -1], inp["f"][:, -1])
    assert np.array_equal(flux[:, -1], inp["flux"][:, -1])


@pytest.mark.parametrize("shape", [(64, 32, 28), (37, 32, 17), (21, 33, 33), (10, 6, 64), (258, 31, 28), (7, 2, 4)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_batch_form_of_the_kernel_on_single_tracers(M, oracle, monkeypatch, shape, variant):
    """The kernel has two forms (mpdata_kernel_wm_body.h): streaming fetch / store for one tracer per
    launch, default cache policy with one instruction per array for tracer batches.
    The NOSTREAM launch switch (mpdata_set_wm_flags) runs the batch form on single-tracer problems, so
    that it sees all the shapes."""
    M.set_wm_flags(M.WMF_NOSTREAM)
    var = M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST
    M.set_variant(var)
    inp = oracle.make_inputs(*shape, seed=13, dist=3)
    f, flux = run_plan_host(M, inp)
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    assert_parity(M, oracle, var, 3, f, flux, f_ref, flux_ref)


@pytest.mark.parametrize("shape", [(64, 32, 28), (37, 32, 17), (21, 33, 33), (10, 6, 64), (130, 31, 12), (7, 2, 4)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("ntr", [2, 3, 7])
@pytest.mark.parametrize("mode", ["exact", "fast", "exact-one-tracer-per-wave", "exact-odd-tracer-in-the-batch",
                                  "exact-odd-tracer-behind-the-batch", "fast-odd-tracer-behind-the-batch"])
def test_tracer_batches_two_tracers_per_wave(M, oracle, monkeypatch, shape, ntr, mode):
    """Tracer batches run TWO tracers per wave (u, w and the tracer-independent factors formed once
    for both); the last tracer of an odd count is taken by ONE more wave per tile of the same launch, through the
    one-tracer form of the body (round 5: mpdata_advect_wm_odd_kernel) -- or, with the SPLIT launch switch, goes
    through the one-tracer kernel behind the batch (rounds 2-4), or, with NOSPLIT, stays in a two-tracer wave
    paired with an empty buffer range.  Every tracer must equal a single-tracer call of the oracle, on all lane mappings
    (nz <= 8 / 16 / 32 / 64) and ragged tile counts.  The TPW1 switch keeps the one-tracer-per-wave batch kernel
    covered."""
    if mode.endswith("per-wave"):
        M.set_wm_flags(M.WMF_TPW1)
    if mode.endswith("in-the-batch"):
        M.set_wm_flags(M.WMF_NOSPLIT)
    if mode.endswith("behind-the-batch"):
        M.set_wm_flags(M.WMF_SPLIT)
    var = M.VARIANT_FAST if mode.startswith("fast") else M.VARIANT_EXACT
    M.set_variant(var)
    ncrms, nx, nz = shape
    base = oracle.make_inputs(ncrms, nx, nz, seed=21, dist=3)
    fs = [oracle.make_inputs(ncrms, nx, nz, seed=210 + t, dist=3)["f"] for t in range(ntr)]
    inp = dict(base)
    inp["f"] = np.asfortranarray(np.stack(fs, axis=-1))
    inp["flux"] = np.asfortranarray(np.stack([base["flux"]] * ntr, axis=-1))
    f, flux = run_plan_host(M, inp, ntr=ntr)
    for t in range(ntr):
        f_ref, flux_ref = oracle.advect(dict(base, f=fs[t].copy()), nthreads=4)
        assert_parity(M, oracle, var, 3, f[..., t], flux[..., t], f_ref, flux_ref)


@pytest.mark.parametrize("shape", [(64, 32, 28), (37, 32, 17), (21, 33, 33), (10, 6, 64), (130, 31, 12), (7, 2, 4), (258, 5, 28)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("ntr", [8, 9, 16])
@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_tracer_batches_whose_workgroups_hold_one_tile(M, oracle, shape, ntr, mode):
    """Tracer pairs per tile a multiple of four (8, 9 = 8 + the odd one, 16 tracers): the four waves of a workgroup are
    then four pairs of the SAME tile (the XCD walk of the batch launch).  Every tracer against the oracle on all lane
    mappings and ragged tile counts."""
    var = M.VARIANT_FAST if mode == "fast" else M.VARIANT_EXACT
    M.set_variant(var)
    ncrms, nx, nz = shape
    base = oracle.make_inputs(ncrms, nx, nz, seed=21, dist=3)
    fs = [oracle.make_inputs(ncrms, nx, nz, seed=310 + t, dist=3)["f"] for t in range(ntr)]
    inp = dict(base)
    inp["f"] = np.asfortranarray(np.stack(fs, axis=-1))
    inp["flux"] = np.asfortranarray(np.stack([base["flux"]] * ntr, axis=-1))
    f, flux = run_plan_host(M, inp, ntr=ntr)
    for t in range(ntr):
        f_ref, flux_ref = oracle.advect(dict(base, f=fs[t].copy()), nthreads=4)
        assert_parity(M, oracle, var, 3, f[..., t], flux[..., t], f_ref, flux_ref)


@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_wavemajor_tracer_batch_and_subranges(M, oracle, variant):
    """T tracers sharing u, w, rho, rhow, adz == T single-tracer calls of the oracle; a sub-range
    run touches only its tracers; device import / export round-trips."""
    import torch
    var = M.VARIANT_EXACT if variant == "exact" else M.VARIANT_FAST
    M.set_variant(var)
    ncrms, nx, nz, T = 96, 32, 28, 5
    base = oracle.make_inputs(ncrms, nx, nz, seed=5, dist=1)
    fs = [oracle.make_inputs(ncrms, nx, nz, seed=50 + t, dist=1)["f"] for t in range(T)]
    refs = [oracle.advect(dict(base, f=fs[t].copy()), nthreads=4) for t in range(T)]
    inp = dict(base)
    inp["f"] = np.asfortranarray(np.stack(fs, axis=-1))
    inp["flux"] = np.asfortranarray(np.stack([base["flux"]] * T, axis=-1))
    f, flux = run_plan_host(M, inp, ntr=T)
    for t in range(T):
        assert_parity(M, oracle, var, 1, f[..., t], flux[..., t], refs[t][0], refs[t][1])
    # device import, run tracers 1..2 only, export
    p = M.Plan(ncrms, nx, nz, T)
    d = {k: to_dev(v) for k, v in inp.items()}
    p.import_device(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["adz"], d["flux"])
    p.run(1, 2)
    fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
    p.export_device(fo, flo)
    p.sync()
    fo, flo = to_host(fo), to_host(flo)
    for t in range(T):
        if t in (1, 2):
            assert_parity(M, oracle, var, 1, fo[..., t], flo[..., t], refs[t][0], refs[t][1])
        else:   # untouched: the import/export round trip is the identity
            assert np.array_equal(fo[..., t], inp["f"][..., t])
            assert np.array_equal(flo[..., t], inp["flux"][..., t])
    p.close()


def test_layouts_agree_bitwise(M, oracle):
    """The same plan calls in the reference layout (x-march kernel) and the wave-major layout:
    EXACT f bit-identical (same arithmetic, same order); FAST equal up to the compiler's choice of
    FMA contractions in the two kernels (reference metric, raw inputs)."""
    inp = oracle.make_inputs(200, 32, 28, seed=3, dist=3)
    for var in (M.VARIANT_EXACT, M.VARIANT_FAST):
        M.set_variant(var)
        f1, fl1 = run_plan_host(M, inp)
        M.set_plan_layout(M.LAYOUT_REFERENCE)
        try:
            ncrms, nxp6, nzm = inp["f"].shape[:3]
            p = M.Plan(ncrms, nxp6 - 6, nzm + 1, 1)
            assert p.layout == M.LAYOUT_REFERENCE
            p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
            p.run(); p.sync()
            f2 = np.empty_like(inp["f"], order="F"); fl2 = np.empty_like(inp["flux"], order="F")
            p.download(f2, fl2)
            p.close()
        finally:
            M.set_plan_layout(M.LAYOUT_WAVEMAJOR)
        if var == M.VARIANT_EXACT:
            assert np.array_equal(f1, f2)
            # flux: the wave-major plan is bit-identical to the reference (parked limited fluxes, round 4); the
            # x-march kernel of a reference-layout plan adds its two partial sums (1e-13), level nz untouched in both
            nzm_ = fl1.shape[1] - 1
            assert np.all(np.abs(fl1[:, :nzm_] - fl2[:, :nzm_]) <= 1e-13 * np.maximum(1.0, np.abs(fl1[:, :nzm_])))
            assert np.array_equal(fl1[:, nzm_], fl2[:, nzm_])
            _, flux_ref = oracle.advect(inp, nthreads=4)
            assert np.array_equal(fl1, flux_ref)
        else:
            assert oracle.rel_l1(f1, f2) < TOL_RELL1
            assert oracle.rel_l1(fl1[:, :-1], fl2[:, :-1]) < TOL_RELL1


def _full_size_check(M, oracle, variant, ncrms, nx, nz, T, blocks, ngpus_devices=None):
    """Device-generated inputs (the counter-based law is shard-invariant), a plan run, then
    (i) sampled blocks of instances against the oracle on regenerated inputs -- instances are
    independent (reference :505-637) -- for EVERY tracer, (ii) the size-independent output
    contract on the whole arrays of every tracer."""
    import torch
    M.set_variant(variant)
    sh = M.shapes(ncrms, nx, nz, 1)
    d = {k: torch.empty(s, dtype=torch.float64, device="cuda:0") for k, s in sh.items()}
    for k in ("u", "w", "rho", "rhow", "adz", "flux"):
        M.fill_synthetic(d[k], k, 100, oracle.DIST_CONDITIONED)
    p = M.Plan(ncrms, nx, nz, T)
    assert p.layout == M.LAYOUT_WAVEMAJOR
    p.import_device(None, d["u"], d["w"], d["rho"], d["rhow"], d["adz"], None)
    for t in range(T):
        M.fill_synthetic(d["f"], "f", 100 + t, oracle.DIST_CONDITIONED)
        p.import_device(d["f"], flux=d["flux"], first_tracer=t)
    p.run()
    p.sync()
    flux_in_top = d["flux"][nz - 1].clone()
    fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
    shared = {}
    for t in range(T):
        p.export_device(fo, flo, first_tracer=t)
        M.fill_synthetic(d["f"], "f", 100 + t, oracle.DIST_CONDITIONED)   # the tracer's input again
        torch.cuda.synchronize()
        # contract: columns -2 / nx+3 and flux level nz untouched, interior >= 0, all finite
        assert torch.equal(fo[:, 0, :], d["f"][:, 0, :]) and torch.equal(fo[:, nx + 5, :], d["f"][:, nx + 5, :])
        assert torch.equal(flo[nz - 1], flux_in_top)
        assert float(fo[:, 3:3 + nx, :].min()) >= 0.0
        assert bool(torch.isfinite(fo).all()) and bool(torch.isfinite(flo).all())
        for s0, n in blocks:
            if (s0, n) not in shared:
                shared[(s0, n)] = oracle.make_inputs(n, nx, nz, seed=100, dist=oracle.DIST_CONDITIONED,
                                                     ncrms_global=ncrms, sl0=s0)
            inp = dict(shared[(s0, n)])
            inp["f"] = oracle.fill_array("f", (n, nx + 6, nz - 1), 100 + t, oracle.DIST_CONDITIONED,
                                         ncrms_global=ncrms, sl0=s0)
            f_ref, flux_ref = oracle.advect(inp)
            f = to_host(fo[..., s0:s0 + n])
            flux = to_host(flo[..., s0:s0 + n])
            if variant == M.VARIANT_EXACT:
                assert np.array_equal(f, f_ref) and flux_close(flux, flux_ref), (t, s0)
            else:
                assert max_abs(f, f_ref) < TOL_ABS and max_abs(flux, flux_ref) < TOL_ABS, (t, s0)
    p.close()


@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_full_size_config3_plan_sampled_against_oracle(M, oracle, variant):
    """BASELINE.json configs[2]: ncrms=65536 nx=32 nz=28 fp64, 1 tracer, through the plan API."""
    _full_size_check(M, oracle, variant, 65536, 32, 28, 1, ((0, 64), (30000 + 7, 50), (65536 - 33, 33)))


@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_full_size_config4_sampled_against_oracle(M, oracle, variant):
    """BASELINE.json configs[3] (= the per-GPU problem of configs[4]): ncrms=65536, 25 tracers
    batched per CRM instance (13.45 GB of f), every tracer checked."""
    _full_size_check(M, oracle, variant, 65536, 32, 28, 25, ((0, 40), (41234 + 5, 31), (65536 - 17, 17)))


def test_plan_arrays_larger_than_4GiB(M, oracle):
    """ncrms = 600000 (f, u, w 4.8-4.9 GB each; the wave-major kernel addresses a tile with 32-bit
    offsets relative to per-wave descriptor bases, so array size does not matter)."""
    _full_size_check(M, oracle, M.VARIANT_EXACT, 600000, 32, 28, 1,
                     ((0, 40), (123456 + 3, 37), (299990, 50), (600000 - 21, 21)))


def test_random_shapes_both_forms(M, oracle, monkeypatch):
    """40 seeded random shapes (ncrms 1..300, nx 1..70, nz 3..64), alternately through the streaming
    form and the batch form of the kernel, EXACT: f bit-identical to the oracle."""
    rng = np.random.default_rng(20261004)
    M.set_variant(M.VARIANT_EXACT)
    for it in range(40):
        ncrms = int(rng.integers(1, 301))
        nx = int(rng.integers(1, 71))
        nz = int(rng.choice([rng.integers(3, 9), rng.integers(9, 17), rng.integers(17, 33), rng.integers(33, 65)]))
        M.set_wm_flags(M.WMF_NOSTREAM if it % 2 else 0)
        inp = oracle.make_inputs(ncrms, nx, nz, seed=1000 + it, dist=3 if it % 3 else 1)
        f, flux = run_plan_host(M, inp)
        f_ref, flux_ref = oracle.advect(inp, nthreads=4)
        assert np.array_equal(f, f_ref), (ncrms, nx, nz, it, max_abs(f, f_ref))
        assert flux_close(flux, flux_ref), (ncrms, nx, nz, it)


def test_random_shapes_tracer_batches(M, oracle):
    """24 seeded random shapes (ncrms 1..200, every nx mod 6 class, nz 3..64) with 2..5 tracers through
    the two-tracers-per-wave kernel (and the one-tracer kernel for an odd last tracer), EXACT: every
    tracer's f bit-identical to a single-tracer call of the oracle."""
    rng = np.random.default_rng(20261005)
    M.set_variant(M.VARIANT_EXACT)
    for it in range(24):
        ncrms = int(rng.integers(1, 201))
        nx = int(rng.integers(1, 9)) * 6 + it % 6 - int(rng.integers(0, 2)) * 6
        nx = max(nx, 1)
        nz = int(rng.choice([rng.integers(3, 9), rng.integers(9, 17), rng.integers(17, 33), rng.integers(33, 65)]))
        ntr = int(rng.integers(2, 6))
        base = oracle.make_inputs(ncrms, nx, nz, seed=3000 + it, dist=3 if it % 3 else 1)
        fs = [oracle.make_inputs(ncrms, nx, nz, seed=4000 + 10 * it + t, dist=3 if it % 3 else 1)["f"] for t in range(ntr)]
        inp = dict(base)
        inp["f"] = np.asfortranarray(np.stack(fs, axis=-1))
        inp["flux"] = np.asfortranarray(np.stack([base["flux"]] * ntr, axis=-1))
        f, flux = run_plan_host(M, inp, ntr=ntr)
        for t in range(ntr):
            f_ref, flux_ref = oracle.advect(dict(base, f=fs[t].copy()), nthreads=4)
            assert np.array_equal(f[..., t], f_ref), (ncrms, nx, nz, ntr, t, it, max_abs(f[..., t], f_ref))
            assert flux_close(np.asfortranarray(flux[..., t]), flux_ref), (ncrms, nx, nz, ntr, t, it)


def test_exact_flux_without_the_park_array(oracle):
    """MPDATA_EXACT_FLUX=sum (read once per process: a child process): EXACT plans without the park array of the
    bit-identical flux -- f still bit-identical, flux(:, 1:nzm) = upwind sum + limited sum, equal to 1e-13 relative."""
    import os, subprocess, sys, json
    code = (
        "import json, numpy as np, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import codesign_kernels_amd as M\n"
        "from oracle import oracle as O\n"
        "O.build_lib(); M.set_variant(M.VARIANT_EXACT)\n"
        "res = {'f': True, 'flux_close': True, 'flux_equal': True}\n"
        "for shape in ((130, 31, 28), (9, 20, 72), (7, 11, 80), (5, 9, 130)):\n"
        "    inp = O.make_inputs(*shape, seed=5, dist=3)\n"
        "    p = M.Plan(*shape, 1); p.upload(inp['f'], inp['u'], inp['w'], inp['rho'], inp['rhow'], inp['adz'], inp['flux']); p.run(); p.sync()\n"
        "    f = np.empty_like(inp['f'], order='F'); fl = np.empty_like(inp['flux'], order='F'); p.download(f, fl); p.close()\n"
        "    fr, flr = O.advect(inp, nthreads=2)\n"
        "    d = np.abs(fl[:, :-1] - flr[:, :-1]); tol = 1e-13 * np.maximum(1.0, np.abs(flr[:, :-1]))\n"
        "    res['f'] &= bool(np.array_equal(f, fr)); res['flux_close'] &= bool(np.all(d <= tol)); res['flux_equal'] &= bool(np.array_equal(fl, flr))\n"
        "print('RESULT ' + json.dumps(res))\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MPDATA_EXACT_FLUX="sum"), capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["f"] and res["flux_close"]


def test_exact_flux_through_the_park_array_everywhere(oracle):
    """MPDATA_EXACT_FLUX=hbm (read once per process: a child process): round 4's park array + finishing kernel in place of the
    register park of round 5 -- the path plans with nx > 36, mpdata_plan_run_uw and 16-wave x-march tilings still take --
    for a plan run, a tracer batch and a reference-layout device call: f AND flux bit-identical."""
    import os, subprocess, sys, json
    code = (
        "import json, numpy as np, sys, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import codesign_kernels_amd as M\n"
        "from oracle import oracle as O\n"
        "from util import run_hip\n"
        "O.build_lib(); M.set_variant(M.VARIANT_EXACT)\n"
        "ok = True\n"
        "for shape, T in (((130, 31, 28), 1), ((37, 9, 17), 3), ((9, 20, 72), 2), ((5, 11, 125), 1)):\n"
        "    base = O.make_inputs(*shape, seed=5, dist=3)\n"
        "    fs = [O.make_inputs(*shape, seed=50 + t, dist=3)['f'] for t in range(T)]\n"
        "    inp = dict(base)\n"
        "    if T > 1:\n"
        "        inp['f'] = np.asfortranarray(np.stack(fs, axis=-1)); inp['flux'] = np.asfortranarray(np.stack([base['flux']] * T, axis=-1))\n"
        "    else:\n"
        "        inp['f'] = fs[0]\n"
        "    p = M.Plan(*shape, T); p.upload(inp['f'], inp['u'], inp['w'], inp['rho'], inp['rhow'], inp['adz'], inp['flux']); p.run(); p.sync()\n"
        "    f = np.empty_like(inp['f'], order='F'); fl = np.empty_like(inp['flux'], order='F'); p.download(f, fl); p.close()\n"
        "    for t in range(T):\n"
        "        fr, flr = O.advect(dict(base, f=fs[t].copy()), nthreads=2)\n"
        "        ok &= bool(np.array_equal(f[..., t] if T > 1 else f, fr)) and bool(np.array_equal(fl[..., t] if T > 1 else fl, flr))\n"
        "    one = dict(base, f=fs[0].copy())\n"
        "    fd, fld = run_hip(M, one)\n"
        "    fr, flr = O.advect(one, nthreads=2)\n"
        "    ok &= bool(np.array_equal(fd, fr)) and bool(np.array_equal(fld, flr))\n"
        "print('RESULT ' + json.dumps({'ok': ok}))\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MPDATA_EXACT_FLUX="hbm"), capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])["ok"]
