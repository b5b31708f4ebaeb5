"""The Fortran drivers of the second and third kernel (codesign-kernels_amd/fortran/bwk_driver.F90, nested_hip.F90): the
reference's programs of atmosphere/ and nested_loops/ with libbwk_hip.so / libnlk_hip.so bound through ISO_C_BINDING in
the place of their OpenACC forms -- the bindings INTEGRATION.md sections 7 and 8 show, compiled and run, and their
results held against the oracles (which are pinned to the reference programs themselves)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "codesign-kernels_amd", "fortran")
BWK = os.path.join(FDIR, "bwk_driver")
NLK = os.path.join(FDIR, "nested_hip")


def test_side_drivers_are_built_and_bind_the_c_abi():
    if not (os.path.exists(BWK) and os.path.exists(NLK)):
        pytest.skip("drivers not built (run __graft_entry__.build())")
    syms = subprocess.run(["nm", "-D", "--undefined-only", BWK], capture_output=True, text=True).stdout
    assert "bwk_biharmonic_wk_scalar" in syms and "bwk_set_variant" in syms and "oracle" not in syms
    syms = subprocess.run(["nm", "-D", "--undefined-only", NLK], capture_output=True, text=True).stdout
    for s in ("nlk_high_order_flux", "nlk_high_order_flux_device", "nlk_set_variant", "hipMalloc", "hipMemcpy"):
        assert s in syms, s
    assert "oracle" not in syms


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_bwk_driver_against_the_oracle(tmp_path, variant):
    from oracle import bwk as B
    B.build_lib()
    nelemd = 3
    inp = B.make_inputs(nelemd)                  # the reference's own generator: what the driver's initialize_data draws
    ref = B.biharmonic(inp)
    reffile, dump = tmp_path / "ref.bin", tmp_path / "out.bin"
    reffile.write_bytes(ref.tobytes(order="F"))
    res = subprocess.run([BWK, str(nelemd), str(variant), str(dump), str(reffile)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    out = np.fromfile(dump, dtype=np.float64).reshape(ref.shape, order="F")
    l2 = float(re.search(r"HIP\s+L2 norm:\s*([0-9.Ee+-]+)", res.stdout).group(1))
    if variant == 0:
        assert np.array_equal(out, ref) and l2 == 0.0
    else:
        assert B.l2norm(out, ref) < 1e-13 and l2 < 1e-13
    sc = float(re.search(r"Self-check \(EXACT against FAST\) L2 norm:\s*([0-9.Ee+-]+)", res.stdout).group(1))
    assert sc < 1e-13 and res.stdout.count("HIP  time:") == 2


@pytest.mark.gpu
@pytest.mark.parametrize("sizes,variant", [((640, 70, 100, 10), 0), ((1000, 333, 37, 6), 0), ((640, 70, 100, 10), 1)])
def test_nested_driver_against_the_oracle(tmp_path, sizes, variant):
    from oracle import nlk as N
    N.build_lib()
    nE, nC, nV, nA = sizes
    nml = tmp_path / "nested.nml"
    nml.write_text(f"&nested_nml\n nIters = 3\n nEdges = {nE}\n nCells = {nC}\n nVertLevels = {nV}\n nAdv = {nA}\n/\n")
    dump = tmp_path / "arrays.bin"
    res = subprocess.run([NLK, str(nml), str(variant), str(dump)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Self-check (host-array call against device-resident call): identical" in res.stdout
    assert re.search(r"beyond errTol: 0\s*$", res.stdout, re.M) and "HIP kernel, 3 iterations" in res.stdout
    raw = dump.read_bytes()
    hdr = np.frombuffer(raw, dtype=np.int32, count=5)
    assert tuple(hdr) == (nE, nC, nV, nV, nA)
    coef = np.frombuffer(raw, dtype=np.float64, count=1, offset=20)[0]
    assert coef == N.coef3rd()                   # the fp32 value of 2.14, as the reference holds it
    off = 28

    def take(dtype, shape):
        nonlocal off
        n = int(np.prod(shape))
        a = np.frombuffer(raw, dtype=dtype, count=n, offset=off).reshape(shape, order="F")
        off += n * np.dtype(dtype).itemsize
        return np.asfortranarray(a)

    inp = {"nAdvCellsForEdge": take(np.int32, (nE,)), "advCellsForEdge": take(np.int32, (nA, nE)),
           "minLevelCell": take(np.int32, (nC,)), "maxLevelCell": take(np.int32, (nC,)),
           "tracerCur": take(np.float64, (nV, nC)), "normalThicknessFlux": take(np.float64, (nV, nE)),
           "advMaskHighOrder": take(np.float64, (nV, nE)), "advCoefs": take(np.float64, (nA, nE)),
           "advCoefs3rd": take(np.float64, (nA, nE)), "coef3rdOrder": coef, "nVertLevels": nV}
    got = take(np.float64, (nV, nE))
    assert off == len(raw)
    # the arrays follow the reference's laws (:58-108)
    assert inp["advCellsForEdge"].min() >= 1 and inp["advCellsForEdge"].max() <= nC
    assert inp["maxLevelCell"].min() >= 3 and inp["maxLevelCell"].max() == nV and np.all(inp["nAdvCellsForEdge"] == nA)
    ref = N.high_order_flux(inp)
    if variant == 0:
        assert np.array_equal(got, ref)
    else:
        err = np.abs(got - ref)
        nz = ref != 0
        err[nz] /= np.abs(ref[nz])
        assert err.max() < 1e-10                 # the program's own tolerance (nested_vars.F90:36)
