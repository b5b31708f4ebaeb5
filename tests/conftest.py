import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # MPDATA_TEST_WATCHDOG=<seconds>: dump every thread's Python stack whenever the run has made no
    # progress for that long (diagnosis of a stall; the run goes on)
    wd = os.environ.get("MPDATA_TEST_WATCHDOG")
    if wd:
        import faulthandler
        faulthandler.dump_traceback_later(float(wd), repeat=True, file=sys.stderr)


@pytest.fixture(scope="session", autouse=True)
def _libraries_built():
    """A fresh checkout has no built artefacts: build the HIP libraries (hipcc cross-compiles
    for gfx950 without a GPU, ~25 s) and the Fortran drivers once, as __graft_entry__.build()
    does.  No-op when they are there or when hipcc is absent (a GPU box runs the shipped .so)."""
    import shutil
    import subprocess
    pkg = os.path.join(ROOT, "codesign-kernels_amd")
    libs = [os.path.join(pkg, n) for n in ("libmpdata_hip.so", "libbwk_hip.so", "libnlk_hip.so")]
    if not all(os.path.exists(p) for p in libs) and shutil.which("hipcc"):
        subprocess.run(["make", "-C", os.path.join(pkg, "csrc"), "-j4"], check=True, stdout=subprocess.DEVNULL)
    if shutil.which("hipcc") and all(os.path.exists(p) for p in libs) and \
            not os.path.exists(os.path.join(ROOT, "tests", "stubs", "libmpdata_hip_fakerccl.so")):
        # the test-only link of the product objects against the recording RCCL stand-in (tests/stubs)
        subprocess.run(["make", "-C", os.path.join(pkg, "csrc"), "fakerccl"], check=True, stdout=subprocess.DEVNULL)
    fdir = os.path.join(pkg, "fortran")
    if shutil.which("amdflang") and all(os.path.exists(p) for p in libs):
        for exe, extra in (("advect", []), ("advect_sp", ["single=1"])):
            if not os.path.exists(os.path.join(fdir, exe)):
                subprocess.run(["make", "-C", fdir, "hip=1", *extra], check=True, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure, oracle/): C restatement of the
    reference's advect_scalar2D_cpu, built on demand with gcc."""
    from oracle import oracle as O
    O.build_lib()
    return O


@pytest.fixture(scope="session")
def mpdata():
    import codesign_kernels_amd as M
    return M
