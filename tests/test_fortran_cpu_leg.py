"""The reference's validation sequence with BOTH legs in one Fortran program (reference :48-58: CPU routine, then the
accelerated routine on the same inputs, then compare(), :679-683): tests/fortran/advect_vs_cpu.F90 calls the oracle's C
restatement of advect_scalar2D_cpu and the product's drop-in `advect_scalar2D` through the ISO_C_BINDING shim, and prints
the reference's own lines.  A TEST program: the only Fortran code of the repo that links the oracle; the product's driver
(codesign-kernels_amd/fortran/test_advect.F90) has no CPU routine because the product has no CPU path."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "codesign-kernels_amd", "fortran")
SRC = os.path.join(ROOT, "tests", "fortran", "advect_vs_cpu.F90")
EXE = os.path.join(ROOT, "tests", "fortran", "advect_vs_cpu")


def _build(oracle):
    lib = oracle.build_lib()
    mods = os.path.join(FDIR, "mod_dp")
    objs = [os.path.join(FDIR, o) for o in ("mpdata_grid.o", "mpdata_hip_mod.o")]
    if not (shutil.which("amdflang") and os.path.isdir(mods) and all(os.path.exists(o) for o in objs)):
        pytest.skip("amdflang or the product driver's objects are missing (run __graft_entry__.build())")
    newest = max(os.path.getmtime(p) for p in [SRC, lib, *objs])
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < newest:
        pkg = os.path.join(ROOT, "codesign-kernels_amd")
        subprocess.run(["amdflang", "-O3", "-I" + mods, "-o", EXE, SRC, *objs, "-L" + pkg, "-lmpdata_hip",
                        "-Wl,-rpath," + pkg, lib, "-Wl,-rpath," + os.path.dirname(lib), "-L/opt/rocm/lib", "-lamdhip64",
                        "-lstdc++", "-lgomp"], check=True, capture_output=True, cwd=os.path.dirname(SRC))
    return EXE


def test_program_builds_and_binds_both_legs(oracle):
    exe = _build(oracle)
    syms = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    assert "mpdata_oracle_advect" in syms and "mpdata_advect_scalar2d" in syms


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dist,variant", [((64, 32, 28), 1, 0), ((48, 32, 58), 2, 0), ((200, 32, 28), 1, 1)])
def test_cpu_leg_and_accelerated_leg_agree_in_one_program(oracle, shape, dist, variant):
    exe = _build(oracle)
    res = subprocess.run([exe, *map(str, shape), str(dist), str(variant)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "CPU Timing:" in res.stdout and "HIP Timing:" in res.stdout
    ef = float(re.search(r"Relative L1 Error - f\s*:\s*([0-9.Ee+-]+)", res.stdout).group(1))
    ex = float(re.search(r"Relative L1 Error - flux\s*:\s*([0-9.Ee+-]+)", res.stdout).group(1))
    dm = float(re.search(r"max abs difference - f\s*:\s*([0-9.Ee+-]+)", res.stdout).group(1))
    if variant == 0:
        assert ef == 0.0 and ex == 0.0 and dm == 0.0      # EXACT: the reference's bits, f and flux
    else:
        assert ef < 1e-14 and ex < 1e-13 and dm < 1e-12   # FAST, conditioned inputs
