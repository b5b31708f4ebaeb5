"""CPU tests: the oracle (oracle/mpdata_oracle.c) against the reference's own
outputs.  The golden vectors under tests/golden/ were produced by the reference
Fortran program itself (tests/golden/make_golden.py); the bar is bit-exact."""
import hashlib

import numpy as np
import pytest

from util import golden_cases, load_golden


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_oracle_matches_reference_golden_bitwise(oracle, case):
    inp = oracle.make_inputs(case["ncrms"], case["nx"], case["nz"], seed=case["seed"], dist=case["dist"])
    # the inputs regenerate to the bytes the fixture was made from
    h = hashlib.sha256(b"".join(inp[k].tobytes(order="F")
                                for k in ("adz", "f", "u", "w", "rho", "rhow", "flux"))).hexdigest()
    assert h == case["inputs_sha256"]
    f_ref, flux_ref = load_golden(case)
    assert hashlib.sha256(f_ref.tobytes(order="F")).hexdigest() == case["f_sha256"]
    f, flux = oracle.advect(inp)
    assert np.array_equal(f, f_ref)
    assert np.array_equal(flux, flux_ref)  # includes the untouched level nz


@pytest.mark.parametrize("case", golden_cases("f32"), ids=lambda c: c["name"])
def test_fp32_oracle_matches_fp32_reference_golden_bitwise(oracle, case):
    """The fp32 build of the oracle against the fp32 build of the reference (rp = IEEE
    single, reference :12-13)."""
    inp = oracle.make_inputs(case["ncrms"], case["nx"], case["nz"], seed=case["seed"],
                             dist=case["dist"], dtype=np.float32)
    h = hashlib.sha256(b"".join(inp[k].tobytes(order="F")
                                for k in ("adz", "f", "u", "w", "rho", "rhow", "flux"))).hexdigest()
    assert h == case["inputs_sha256"]
    f_ref, flux_ref = load_golden(case)
    assert f_ref.dtype == np.float32
    f, flux = oracle.advect(inp)
    assert f.dtype == np.float32
    assert np.array_equal(f, f_ref) and np.array_equal(flux, flux_ref)
    # and the fp64 result rounds to within fp32 rounding noise of it (conditioned law only)
    if case["dist"] == 1:
        inp64 = {k: v.astype(np.float64) for k, v in inp.items()}
        f64, _ = oracle.advect({k: np.asfortranarray(v) for k, v in inp64.items()})
        assert np.max(np.abs(f64 - f)) < 2e-5


@pytest.mark.parametrize("case", golden_cases()[:5], ids=lambda c: c["name"])
def test_reference_binary_still_agrees(oracle, case):
    """When oracle/_ref holds the reference executable for a shape, run it."""
    if oracle.ref_exe(case["ncrms"], case["nx"], case["nz"]) is None:
        pytest.skip("no oracle/_ref binary for this shape")
    inp = oracle.make_inputs(case["ncrms"], case["nx"], case["nz"], seed=case["seed"], dist=case["dist"])
    f_ref, flux_ref, _ = oracle.run_reference(inp)
    f, flux = oracle.advect(inp)
    assert np.array_equal(f, f_ref) and np.array_equal(flux, flux_ref)


def test_generators_agree_bitwise(oracle):
    for dist in (1, 2, 3):
        sh = oracle.shapes(5, 8, 6)
        for name, shape in sh.items():
            a = oracle.fill_array(name, shape, 1234, dist)
            b = oracle.fill_array_c(name, shape, 1234, dist)
            assert np.array_equal(a, b), (name, dist)
            a32 = oracle.fill_array(name, shape, 1234, dist, dtype=np.float32)
            b32 = oracle.fill_array_c(name, shape, 1234, dist, dtype=np.float32)
            assert np.array_equal(a32, b32) and np.array_equal(a32, a.astype(np.float32))
    # value ranges of the conditioned law (SURVEY.md 8d "D1")
    inp = oracle.make_inputs(16, 8, 6, seed=3, dist=oracle.DIST_CONDITIONED)
    assert 0 <= inp["f"].min() and inp["f"].max() < 1
    assert -0.5 <= inp["u"].min() and inp["u"].max() < 0.5 and inp["u"].min() < 0
    assert 0.5 <= inp["rho"].min() and inp["adz"].max() < 1.5


def test_generator_is_shard_invariant(oracle):
    full = oracle.make_inputs(12, 8, 6, seed=9, dist=1)
    part = oracle.make_inputs(5, 8, 6, seed=9, dist=1, ncrms_global=12, sl0=4)
    for k in full:
        assert np.array_equal(full[k][4:9], part[k]), k


def test_openmp_chunks_equal_serial(oracle):
    inp = oracle.make_inputs(200, 16, 10, seed=5, dist=3)
    f1, x1 = oracle.advect(inp, nthreads=1)
    f2, x2 = oracle.advect(inp, nthreads=4)
    assert np.array_equal(f1, f2) and np.array_equal(x1, x2)


def test_output_contract(oracle):
    """Reference output contract: halo columns -2, nx+3
    unchanged; flux level nz untouched; w level nz, rhow level nz and the
    incoming flux never read."""
    nx, nz = 8, 7
    inp = oracle.make_inputs(6, nx, nz, seed=11, dist=1)
    f, flux = oracle.advect(inp)
    assert np.array_equal(f[:, 0, :], inp["f"][:, 0, :])
    assert np.array_equal(f[:, nx + 5, :], inp["f"][:, nx + 5, :])
    assert not np.array_equal(f[:, 1, :], inp["f"][:, 1, :])
    assert np.array_equal(flux[:, nz - 1], inp["flux"][:, nz - 1])
    pert = {k: v.copy(order="F") for k, v in inp.items()}
    pert["w"][:, :, nz - 1] += 1.0
    pert["rhow"][:, nz - 1] += 1.0
    pert["flux"][:, : nz - 1] += 1.0
    f2, flux2 = oracle.advect(pert)
    assert np.array_equal(f, f2) and np.array_equal(flux[:, : nz - 1], flux2[:, : nz - 1])
    # interior is positive definite (:634)
    assert f[:, 3:3 + nx, :].min() >= 0.0


def test_instances_are_independent(oracle):
    """No statement couples different sl: the basis of sharding."""
    inp = oracle.make_inputs(9, 8, 6, seed=2, dist=1)
    f, flux = oracle.advect(inp)
    sub = {k: np.asfortranarray(v[3:7]) for k, v in inp.items()}
    fs, xs = oracle.advect(sub)
    assert np.array_equal(f[3:7], fs) and np.array_equal(flux[3:7], xs)


def test_tracer_batch_is_per_tracer_call(oracle):
    inp = oracle.make_inputs(7, 8, 6, seed=4, dist=1, ntracers=3)
    f, flux = oracle.advect(inp)
    for t in range(3):
        one = dict(inp)
        one["f"] = np.asfortranarray(inp["f"][..., t])
        one["flux"] = np.asfortranarray(inp["flux"][..., t])
        ft, xt = oracle.advect(one)
        assert np.array_equal(f[..., t], ft) and np.array_equal(flux[..., t], xt)


def test_bad_sizes_rejected(oracle):
    import ctypes
    z = (ctypes.c_double * 1)()
    assert oracle.lib().mpdata_oracle_advect(4, 8, 2, z, z, z, z, z, z, z, 1) == -1
