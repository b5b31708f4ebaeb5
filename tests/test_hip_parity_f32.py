"""GPU parity tests of the fp32 entry points (reference precision switch `rp`, reference
:12-13; see include/mpdata_hip.h section 6) against the fp32 build of the oracle, which is
pinned bit-for-bit to an fp32 build of the reference (tests/golden/*_f32.npz).

Two kernel families: tile ids 40-43 (two adjacent instances per lane, packed fp32
arithmetic; even ncrms, nz <= 64) and 30-32 (one instance per lane; any ncrms, nz <= 32).

Bars:
  * variant EXACT: f BIT-IDENTICAL to the fp32 oracle / the fp32 reference golden outputs;
    flux to 2e-5 relative (the kernel adds sum(upwind) + sum(limited), the reference adds
    the limited terms one by one onto the finished upwind sum; fp32 sums of 2*nx terms);
  * variant FAST (FMA contraction, Newton reciprocal): max|d| < 1e-5 on the conditioned
    law (outputs O(1); measured 6e-7 = 10 units in the last place; the fp64 bar of 1e-12
    scaled by the ratio of the unit roundoffs would be 5e-4), relative L1 (reference
    :681-682) < 2e-6 on the raw laws (measured 2.3e-7).
"""
import numpy as np
import pytest

from util import golden_cases, load_golden, max_abs, run_hip, to_host

pytestmark = pytest.mark.gpu

F32 = np.float32
TOL_ABS = 1e-5
TOL_RELL1 = 2e-6
SCALAR_TILES = [30, 31, 32]       # LPS 8 / 16 / 32 -> nz <= 8 / 16 / 32
PACKED_TILES = [40, 41, 42, 43]   # LPS 8 / 16 / 32 / 64


def flux_close(flux, flux_ref):
    """EXACT: flux BIT-IDENTICAL to the fp32 build of the reference since round 4 (parked limited fluxes, added in the
    reference's order), level nz untouched."""
    return bool(np.array_equal(flux, flux_ref))


@pytest.fixture(scope="module")
def M(mpdata):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    yield mpdata
    mpdata.set_tile(-1)
    mpdata.set_variant(mpdata.VARIANT_EXACT)


def check(M, oracle, inp, variant, dist):
    assert inp["f"].dtype == F32
    M.set_variant(variant)
    f, flux = run_hip(M, inp)
    assert f.dtype == F32 and flux.dtype == F32
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    if f.ndim == 4:  # tracer batch: flux per tracer
        fl = [(np.asfortranarray(flux[..., t]), np.asfortranarray(flux_ref[..., t])) for t in range(f.shape[3])]
    else:
        fl = [(flux, flux_ref)]
    if variant == M.VARIANT_EXACT:
        assert np.array_equal(f, f_ref), f"f differs: max|d|={max_abs(f, f_ref):.3e}"
        for a, b in fl:
            assert flux_close(a, b), f"flux differs: max|d|={max_abs(a, b):.3e}"
    elif dist == oracle.DIST_CONDITIONED:
        assert max_abs(f, f_ref) < TOL_ABS
        assert max_abs(flux, flux_ref) < TOL_ABS
    else:
        assert oracle.rel_l1(f, f_ref) < TOL_RELL1
        for a, b in fl:
            nzm = a.shape[1] - 1
            assert oracle.rel_l1(a[:, :nzm], b[:, :nzm]) < TOL_RELL1
            assert np.array_equal(a[:, nzm], b[:, nzm])


@pytest.mark.parametrize("case", golden_cases("f32"), ids=lambda c: c["name"])
def test_exact_variant_reproduces_fp32_reference_golden_bitwise(M, oracle, case):
    """HIP vs the outputs of the fp32 build of the reference program itself; every kernel
    family that covers the shape."""
    M.set_variant(M.VARIANT_EXACT)
    inp = oracle.make_inputs(case["ncrms"], case["nx"], case["nz"], seed=case["seed"],
                             dist=case["dist"], dtype=F32)
    f_ref, flux_ref = load_golden(case)
    tiles = [-1] + [t for t, lps in zip(SCALAR_TILES, (8, 16, 32)) if case["nz"] <= lps][:1]
    if case["ncrms"] % 2 == 0:
        tiles += [t for t, lps in zip(PACKED_TILES, (8, 16, 32, 64)) if case["nz"] <= lps][:1]
    try:
        for tile in tiles:
            M.set_tile(tile)
            f, flux = run_hip(M, inp)
            assert np.array_equal(f, f_ref), f"tile {tile}: max|df|={max_abs(f, f_ref):.3e}"
            assert flux_close(flux, flux_ref), f"tile {tile}: max|dflux|={max_abs(flux, flux_ref):.3e}"
    finally:
        M.set_tile(-1)


@pytest.mark.parametrize("tile", [32, 42, 43])
@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_fp32_tilings_config1_and_config2(M, oracle, tile, variant):
    """BASELINE.json configs[0] (ncrms=64) and configs[1] (ncrms=4096), nx=32 nz=28, in fp32."""
    M.set_tile(tile)
    try:
        for ncrms in (64, 4096):
            inp = oracle.make_inputs(ncrms, 32, 28, seed=100, dist=oracle.DIST_CONDITIONED, dtype=F32)
            check(M, oracle, inp, variant, oracle.DIST_CONDITIONED)
    finally:
        M.set_tile(-1)


@pytest.mark.parametrize("shape", [(37, 32, 28), (38, 32, 28), (1, 32, 28), (2, 32, 28), (100, 8, 6),
                                   (130, 1, 3), (17, 5, 3), (48, 32, 58), (34, 20, 4), (70, 60, 9),
                                   (20, 130, 5), (19, 300, 8), (22, 12, 16), (18, 7, 17), (35, 9, 32),
                                   (36, 9, 32), (10, 6, 33), (6, 4, 64)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_fp32_ragged_and_edge_shapes(M, oracle, shape, variant):
    """Odd ncrms (one-instance-per-lane kernels) and even ncrms (packed kernels), ncrms not a
    multiple of the tile, minimum nz, every lanes-per-instance boundary, nx from 1 to 300."""
    M.set_tile(-1)
    for dist in (oracle.DIST_CONDITIONED, oracle.DIST_RAW_SIGNED):
        inp = oracle.make_inputs(*shape, seed=21, dist=dist, dtype=F32)
        check(M, oracle, inp, variant, dist)


def test_fp32_unsupported_shapes_fail_loudly(M, oracle):
    """Odd ncrms needs nz <= 32; nz > 238 has no fp32 kernel (65 .. 238 with an even ncrms: through the calling thread's
    wave-major plan since round 5): an error, never a fallback."""
    for shape in ((9, 6, 33), (5, 10, 65), (6, 10, 239)):
        inp = oracle.make_inputs(*shape, seed=3, dist=1, dtype=F32)
        with pytest.raises(M.MpdataError) as ei:
            run_hip(M, inp)
        assert ei.value.code == -2


@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_fp32_many_small_shapes(M, oracle, variant):
    rng = np.random.default_rng(4321)
    M.set_tile(-1)
    shapes = [(int(rng.integers(1, 70)), nx, int(rng.integers(3, 20))) for nx in range(1, 19)]
    shapes += [(int(rng.integers(1, 40)), int(rng.integers(1, 40)), nz) for nz in (3, 7, 8, 9, 15, 16, 17, 31, 32)]
    for shape in shapes:
        inp = oracle.make_inputs(*shape, seed=int(rng.integers(1, 10**6)), dist=oracle.DIST_CONDITIONED, dtype=F32)
        check(M, oracle, inp, variant, oracle.DIST_CONDITIONED)


@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_fp32_reference_raw_distribution(M, oracle, variant):
    inp = oracle.make_inputs(256, 32, 28, seed=100, dist=oracle.DIST_RAW, dtype=F32)
    check(M, oracle, inp, variant, oracle.DIST_RAW)


@pytest.mark.parametrize("ncrms", [96, 97])
@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_fp32_tracer_batch(M, oracle, variant, ncrms):
    inp = oracle.make_inputs(ncrms, 32, 28, seed=8, dist=oracle.DIST_CONDITIONED, ntracers=5, dtype=F32)
    check(M, oracle, inp, variant, oracle.DIST_CONDITIONED)


def test_fp32_host_dropin_call(M, oracle):
    M.set_variant(M.VARIANT_EXACT)
    inp = oracle.make_inputs(150, 32, 28, seed=31, dist=oracle.DIST_CONDITIONED, dtype=F32)
    f_ref, flux_ref = oracle.advect(inp)
    f = inp["f"].copy(order="F")
    flux = inp["flux"].copy(order="F")
    M.advect_scalar2D_host(f, inp["u"], inp["w"], inp["rho"], inp["rhow"], flux, inp["adz"])
    assert np.array_equal(f, f_ref) and flux_close(flux, flux_ref)
    plan = M.Plan(150, 32, 28, dtype=F32)
    plan.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    plan.run()
    plan.sync()
    assert plan.last_kernel_ms() > 0
    f2, flux2 = np.empty_like(f), np.empty_like(flux)
    plan.download(f2, flux2)
    plan.close()
    # (an fp32 PLAN with an even ncrms runs the wave-major kernels: flux bit-identical too since round 4)
    assert np.array_equal(f2, f_ref) and np.array_equal(flux2, flux_ref)
    with pytest.raises(M.MpdataError):  # mixed precisions are refused
        M.advect_scalar2D_host(f, inp["u"].astype(np.float64, order="F"), inp["w"], inp["rho"],
                               inp["rhow"], flux, inp["adz"])


def test_fp32_device_generator_matches_numpy(M, oracle):
    import torch
    for dist in (1, 2, 3):
        sh = M.shapes(40, 8, 6)
        for name, shape in sh.items():
            t = torch.empty(shape, dtype=torch.float32, device="cuda:0")
            M.fill_synthetic(t, name, 77, dist, ncrms_global=100, sl0=13)
            ref = oracle.fill_array(name, tuple(reversed(shape)), 77, dist, ncrms_global=100, sl0=13, dtype=F32)
            assert np.array_equal(to_host(t), ref), (name, dist)


@pytest.mark.parametrize("variant", [0, 1], ids=["exact", "fast"])
def test_fp32_full_size_config3_sampled_against_oracle(M, oracle, variant):
    """BASELINE.json configs[2] in fp32: ncrms=65536 nx=32 nz=28, generated on the device;
    blocks of instances against the oracle plus the output contract on the whole arrays."""
    import torch
    ncrms, nx, nz = 65536, 32, 28
    M.set_variant(variant)
    M.set_tile(-1)
    sh = M.shapes(ncrms, nx, nz)
    d = {k: torch.empty(s, dtype=torch.float32, device="cuda:0") for k, s in sh.items()}
    for k in d:
        M.fill_synthetic(d[k], k, 100, oracle.DIST_CONDITIONED)
    f0_lo = d["f"][:, 0, :].clone()
    f0_hi = d["f"][:, nx + 5, :].clone()
    flux_top = d["flux"][nz - 1].clone()
    M.advect_scalar2D(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"])
    torch.cuda.synchronize()
    assert torch.equal(d["f"][:, 0, :], f0_lo) and torch.equal(d["f"][:, nx + 5, :], f0_hi)
    assert torch.equal(d["flux"][nz - 1], flux_top)
    assert float(d["f"][:, 3:3 + nx, :].min()) >= 0.0
    assert bool(torch.isfinite(d["f"]).all()) and bool(torch.isfinite(d["flux"]).all())
    for s0, n in ((0, 64), (30000 + 7, 50), (ncrms - 33, 33)):
        inp = oracle.make_inputs(n, nx, nz, seed=100, dist=oracle.DIST_CONDITIONED,
                                 ncrms_global=ncrms, sl0=s0, dtype=F32)
        f_ref, flux_ref = oracle.advect(inp)
        f = to_host(d["f"][..., s0:s0 + n])
        flux = to_host(d["flux"][..., s0:s0 + n])
        if variant == M.VARIANT_EXACT:
            assert np.array_equal(f, f_ref) and flux_close(flux, flux_ref)
        else:
            assert max_abs(f, f_ref) < TOL_ABS and max_abs(flux, flux_ref) < TOL_ABS


@pytest.mark.parametrize("shape", [(64, 32, 28), (48, 32, 58), (2, 1, 3), (6, 3, 8), (34, 7, 16), (130, 9, 32), (22, 33, 33),
                                   (10, 6, 64), (258, 31, 28)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_fp32_plan_wavemajor_shapes(M, oracle, shape, variant):
    """fp32 plans with an even ncrms live in the wave-major layout too (two adjacent instances per
    lane = 8-byte elements, packed fp32 arithmetic, mpdata_kernel_wm_body.h): EXACT bit-identical in f
    to the fp32 oracle, FAST within the fp32 tolerances of this file; odd ncrms keeps the reference
    layout."""
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
        assert np.array_equal(f, f_ref)
    else:
        assert np.abs(f.astype(np.float64) - f_ref).max() < 1e-5
    d = np.abs(flux[:, :nzm].astype(np.float64) - flux_ref[:, :nzm])
    assert np.all(d <= 2e-5 * np.maximum(1.0, np.abs(flux_ref[:, :nzm])))
    assert np.array_equal(flux[:, nzm], inp["flux"][:, nzm])
    if shape[2] <= 32:   # odd ncrms: one instance per lane, reference layout (nz <= 32 only)
        q = M.Plan(shape[0] + 1, shape[1], shape[2], 1, dtype=F32)
        assert q.layout == M.LAYOUT_REFERENCE
        q.close()
