"""Several host threads, each driving its OWN plans on its OWN stream at the same time (ctypes releases the interpreter
lock in every call: the threads really are inside the library together).  The library keeps no per-call state outside a
plan (buffers belong to the plan, the error text to the thread), so every thread must get the oracle's result bit for
bit (EXACT), for the plan API, for mpdata_plan_run_uw and for the reference-layout device call (whose EXACT form
allocates and frees its park array in stream order)."""
import threading

import numpy as np
import pytest

from util import to_dev, to_host

pytestmark = pytest.mark.gpu


def test_threads_with_their_own_plans_and_streams(mpdata, oracle):
    import torch
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    M.set_wm_flags(0)
    # (72 levels: several waves per instance in the plan, and the device call goes through a plan the library keeps for
    #  the calling THREAD -- freed when the thread ends)
    shapes = [(96, 32, 28), (50, 17, 12), (64, 32, 58), (33, 9, 20), (34, 9, 72)]
    nthreads, rounds = 5, 6
    cases = []
    for t, (ncrms, nx, nz) in enumerate(shapes):
        inp = oracle.make_inputs(ncrms, nx, nz, seed=4100 + t, dist=1)
        f_ref, flux_ref = oracle.advect(inp)
        f2_ref, _ = oracle.advect(dict(inp, f=f_ref))       # a second step on the same velocities
        cases.append((ncrms, nx, nz, inp, f_ref, flux_ref, f2_ref))
    errors = []
    start = threading.Barrier(nthreads)

    def work(t):
        try:
            ncrms, nx, nz, inp, f_ref, flux_ref, f2_ref = cases[t]
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                d = {k: to_dev(v) for k, v in inp.items()}
                s.synchronize()
                start.wait()
                for r in range(rounds):
                    # plan API: two steps on the plan's own state
                    p = M.Plan(ncrms, nx, nz, 1)
                    p.set_stream()
                    p.import_device(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["adz"], d["flux"])
                    p.run()
                    f1 = torch.empty_like(d["f"]); fl1 = torch.empty_like(d["flux"])
                    p.export_device(f1, fl1)
                    p.run()
                    f2 = torch.empty_like(d["f"]); fl2 = torch.empty_like(d["flux"])
                    p.export_device(f2, fl2)
                    # a step on fresh reference-layout velocities from the first state
                    p.import_device(d["f"], None, None, None, None, None, None)
                    p.run_uw(d["u"], d["w"])
                    f3 = torch.empty_like(d["f"]); fl3 = torch.empty_like(d["flux"])
                    p.export_device(f3, fl3)
                    p.close()
                    # the reference-layout device call on this thread's stream
                    f4 = d["f"].clone(); fl4 = d["flux"].clone()
                    M.advect_scalar2D(f4, d["u"], d["w"], d["rho"], d["rhow"], fl4, d["adz"])
                    s.synchronize()
                    nzm = nz - 1
                    for name, f, fl in (("plan", f1, fl1), ("run_uw", f3, fl3), ("device call", f4, fl4)):
                        assert np.array_equal(to_host(f), f_ref), (t, r, name, "f")
                        assert np.array_equal(to_host(fl)[:, :nzm], flux_ref[:, :nzm]), (t, r, name, "flux")
                    assert np.array_equal(to_host(f2), f2_ref), (t, r, "second step")
        except BaseException as exc:   # (assertions included: reported by the main thread)
            errors.append((t, repr(exc)))
            try:
                start.abort()
            except Exception:
                pass

    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not errors, errors
    assert not any(x.is_alive() for x in th)
