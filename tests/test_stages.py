"""Stage-by-stage debug mode (include/mpdata_hip.h section 7): the routine
as eight unfused kernels that materialise the reference's temporaries, stoppable after any
stage.  Every array after every stage must equal the oracle's bit for bit; the final state
must equal the fused kernels'.  The CPU part checks the oracle's own stage mode."""
import numpy as np
import pytest

from util import run_hip, to_dev, to_host


def test_oracle_stage_mode_is_consistent(oracle):
    """Stage 8 of the stage mode == the plain routine; earlier stops leave later outputs alone."""
    inp = oracle.make_inputs(7, 9, 6, seed=17, dist=oracle.DIST_RAW_SIGNED)
    f, flux = oracle.advect(inp)
    s8 = oracle.advect_stages(inp, 8)
    assert np.array_equal(s8["f"], f) and np.array_equal(s8["flux"], flux)
    s2 = oracle.advect_stages(inp, 2)
    assert np.array_equal(s2["f"], inp["f"])                      # f is first touched in stage 3
    assert not np.array_equal(s2["flux"][:, :-1], inp["flux"][:, :-1])
    s3 = oracle.advect_stages(inp, 3)
    assert np.array_equal(s3["f"][:, 0], inp["f"][:, 0]) and not np.array_equal(s3["f"][:, 1], inp["f"][:, 1])
    assert np.all(s3["www"][:, :, -1] == 0.0)                     # www(:,:,nz) = 0 (:511)
    s4 = oracle.advect_stages(inp, 4)
    assert np.all(s4["www"][:, :, 0] == 0.0)                      # www(:,:,1) = 0 (:586)
    s6 = oracle.advect_stages(inp, 6)
    assert np.all(s6["mx"] >= 0.0) and np.all(s6["mn"] >= 0.0)    # limiter ratios are non-negative


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dist", [((70, 32, 28), 1), ((33, 9, 6), 3), ((5, 1, 3), 2), ((130, 40, 17), 1)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else f"dist{v}")
def test_every_stage_matches_the_oracle_bitwise(mpdata, oracle, shape, dist):
    import torch
    M = mpdata
    ncrms, nx, nz = shape
    inp = oracle.make_inputs(ncrms, nx, nz, seed=23, dist=dist)
    FILL = -7.25   # parts of uuu / www / mx / mn are never written (as in the reference)
    for last in range(1, 9):
        ref = oracle.advect_stages(inp, last, fill=FILL)
        d = {k: to_dev(v) for k, v in inp.items()}
        tmp = {k: torch.full(s, FILL, dtype=torch.float64, device="cuda:0") for k, s in M.stage_shapes(ncrms, nx, nz).items()}
        M.debug_stages(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"], tmp, last)
        torch.cuda.synchronize()
        got = {"f": to_host(d["f"]), "flux": to_host(d["flux"]), **{k: to_host(v) for k, v in tmp.items()}}
        for k in ("f", "flux", "uuu", "www", "mx", "mn"):
            assert np.array_equal(got[k], ref[k]), f"stage {last}: {k} differs, max|d|={np.abs(got[k] - ref[k]).max():.3e}"


@pytest.mark.gpu
def test_stage_mode_agrees_with_the_fused_kernels(mpdata, oracle):
    """All eight stages == the fused EXACT kernels (f bitwise; flux bitwise against the
    k-marching kernel, which keeps the reference's summation order)."""
    import torch
    M = mpdata
    inp = oracle.make_inputs(96, 32, 28, seed=5, dist=oracle.DIST_CONDITIONED)
    d = {k: to_dev(v) for k, v in inp.items()}
    tmp = {k: torch.zeros(s, dtype=torch.float64, device="cuda:0") for k, s in M.stage_shapes(96, 32, 28).items()}
    M.debug_stages(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"], tmp, 8)
    torch.cuda.synchronize()
    f_st, flux_st = to_host(d["f"]), to_host(d["flux"])
    M.set_variant(M.VARIANT_EXACT)
    f_x, _ = run_hip(M, inp)
    M.set_tile(0)
    try:
        f_k, flux_k = run_hip(M, inp)
    finally:
        M.set_tile(-1)
    assert np.array_equal(f_st, f_x) and np.array_equal(f_st, f_k)
    assert np.array_equal(flux_st, flux_k)


@pytest.mark.gpu
def test_stage_mode_rejects_bad_arguments(mpdata, oracle):
    import torch
    M = mpdata
    inp = oracle.make_inputs(4, 8, 6, seed=1, dist=1)
    d = {k: to_dev(v) for k, v in inp.items()}
    tmp = {k: torch.zeros(s, dtype=torch.float64, device="cuda:0") for k, s in M.stage_shapes(4, 8, 6).items()}
    for bad in (0, 9):
        with pytest.raises(M.MpdataError):
            M.debug_stages(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"], tmp, bad)
    tmp["mx"] = tmp["mx"][:-1].contiguous()
    with pytest.raises(M.MpdataError):
        M.debug_stages(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"], tmp, 8)
