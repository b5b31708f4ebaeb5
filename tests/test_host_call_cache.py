"""The host-array call keeps its streams and chunk buffers per host thread between calls (creating and destroying them
cost 6.7 ms per call: mpdata_hostcall.hip HostCtx).  Calls of growing and shrinking size, of several tracer counts and from
several threads must each give the oracle's result bit for bit; releasing the buffers, and MPDATA_HOST_CACHE=0, must
change nothing but the time."""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _call(M, oracle, ncrms, nx, nz, seed, ntr=1):
    inp = oracle.make_inputs(ncrms, nx, nz, seed=seed, dist=1)
    f_ref, flux_ref = oracle.advect(inp)
    f, flux = inp["f"].copy(order="F"), inp["flux"].copy(order="F")
    M.advect_scalar2D_host(f, inp["u"], inp["w"], inp["rho"], inp["rhow"], flux, inp["adz"])
    assert np.array_equal(f, f_ref) and np.array_equal(flux, flux_ref), (ncrms, nx, nz)


def test_sizes_grow_and_shrink_between_calls(mpdata, oracle):
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    for i, shape in enumerate([(48, 32, 58), (3000, 32, 28), (64, 32, 28), (5000, 16, 12), (48, 32, 58), (2100, 8, 33)]):
        _call(M, oracle, *shape, seed=7000 + i)
    M.release_host_buffers()
    _call(M, oracle, 100, 32, 28, seed=7100)
    M.release_host_buffers()
    M.release_host_buffers()   # (nothing to release: no error)


def test_the_second_call_does_not_pay_for_streams_and_buffers_again(mpdata, oracle):
    M = mpdata
    M.set_variant(M.VARIANT_FAST)
    inp = oracle.make_inputs(48, 32, 58, seed=7200, dist=1)
    ts = []
    for _ in range(6):
        f, flux = inp["f"].copy(order="F"), inp["flux"].copy(order="F")
        t0 = time.perf_counter()
        M.advect_scalar2D_host(f, inp["u"], inp["w"], inp["rho"], inp["rhow"], flux, inp["adz"])
        ts.append(time.perf_counter() - t0)
    M.set_variant(M.VARIANT_EXACT)
    # round 3: 7.5-8.8 ms per call at the reference's shipped size, 95 % of it stream creation and destruction
    assert min(ts[1:]) < 3e-3, ts


def test_threads_keep_their_own_buffers(mpdata, oracle):
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    errors = []

    def work(t):
        try:
            for r in range(3):
                _call(M, oracle, 200 + 300 * t + 64 * r, 16 + t, 12 + 5 * t, seed=7300 + 10 * t + r)
            M.release_host_buffers()
        except BaseException as exc:
            errors.append((t, repr(exc)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not errors, errors


def test_without_the_cache(tmp_path):
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, codesign_kernels_amd as M\n"
            "from oracle import oracle as O\n"
            "O.build_lib()\n"
            "for i, n in enumerate((48, 700, 48)):\n"
            "    inp = O.make_inputs(n, 32, 28, seed=7400 + i, dist=1)\n"
            "    fr, xr = O.advect(inp)\n"
            "    f, x = inp['f'].copy(order='F'), inp['flux'].copy(order='F')\n"
            "    M.advect_scalar2D_host(f, inp['u'], inp['w'], inp['rho'], inp['rhow'], x, inp['adz'])\n"
            "    assert np.array_equal(f, fr) and np.array_equal(x, xr)\n"
            "print('OK')\n") % (ROOT, os.path.join(ROOT, "tests"))
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MPDATA_HOST_CACHE="0", MPDATA_VARIANT="exact"),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "OK" in res.stdout, res.stdout + res.stderr
