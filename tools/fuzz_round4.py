#!/usr/bin/env python3
"""One-off wider sweep than the test suite over round 4's new kernels (GPU):
  * mpdata_plan_run_uw, one tracer, EXACT bitwise vs the oracle: random shapes with nz up to 64 (the one-instance-per-wave
    form, a 16-wave workgroup), even / odd ncrms, nx 1..60, the three input laws;
  * mpdata_plan_run_uw on tracer batches (2..7 tracers): EXACT bitwise per tracer (the converting kernel + the batch kernel
    for nz <= 32 and even ncrms, one conversion pass otherwise), FAST within 1e-12 on conditioned inputs;
  * layout import by row segments (wm_import_rows_kernel) -> export: the round trip of f is the identity, bit for bit, and
    a run after it equals a run after the host upload path (which converts through the same kernels from a staging copy);
  * fp32 plans (two instances per lane: the import sees pairs of floats as 8-byte elements): EXACT bitwise vs the fp32 oracle.
usage: python tools/fuzz_round4.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import codesign_kernels_amd as M
from oracle import oracle as O
from util import to_dev, to_host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
O.build_lib()
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "4242")))   # (shapes; FUZZ_SEED draws another sweep)
bad = 0


def rshape(nzmax=64, ncmax=400, even=None):
    ncrms = int(rng.integers(1, ncmax))
    if even is True:
        ncrms = 2 * max(1, ncrms // 2)
    nz = int(rng.choice([rng.integers(3, 9), rng.integers(9, 17), rng.integers(17, 33), rng.integers(33, nzmax + 1)]))
    return ncrms, int(rng.integers(1, 61)), nz


M.set_variant(M.VARIANT_EXACT)
for it in range(n):
    ncrms, nx, nz = rshape()
    dist = int(rng.integers(1, 4))
    inp = O.make_inputs(ncrms, nx, nz, seed=5000 + it, dist=dist)
    other = O.make_inputs(ncrms, nx, nz, seed=9000 + it, dist=dist)
    p = M.Plan(ncrms, nx, nz, 1)
    p.upload(inp["f"], other["u"], other["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run_uw(to_dev(inp["u"]), to_dev(inp["w"])); p.sync()
    f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
    p.download(f, fl); p.close()
    f_ref, fl_ref = O.advect(inp, nthreads=4)
    if not np.array_equal(f, f_ref):
        bad += 1; print("run_uw MISMATCH", ncrms, nx, nz, dist, np.abs(f - f_ref).max())
print("run_uw one-tracer cases", n, "bad", bad, flush=True)

for it in range(n // 2):
    ncrms, nx, nz = rshape(ncmax=300)
    T = int(rng.integers(2, 8))
    base = O.make_inputs(ncrms, nx, nz, seed=100 + it, dist=1)
    other = O.make_inputs(ncrms, nx, nz, seed=7100 + it, dist=1)
    fs = [O.make_inputs(ncrms, nx, nz, seed=20000 + 10 * it + t, dist=1)["f"] for t in range(T)]
    up = dict(base, u=other["u"], w=other["w"])
    up["f"] = np.asfortranarray(np.stack(fs, axis=-1)); up["flux"] = np.asfortranarray(np.stack([base["flux"]] * T, axis=-1))
    refs = [O.advect(dict(base, f=fs[t].copy()), nthreads=4)[0] for t in range(T)]
    for var in (M.VARIANT_EXACT, M.VARIANT_FAST):
        M.set_variant(var)
        p = M.Plan(ncrms, nx, nz, T)
        p.upload(up["f"], up["u"], up["w"], up["rho"], up["rhow"], up["adz"], up["flux"])
        p.run_uw(to_dev(base["u"]), to_dev(base["w"])); p.sync()
        f = np.empty_like(up["f"], order="F"); fl = np.empty_like(up["flux"], order="F")
        p.download(f, fl); p.close()
        for t in range(T):
            ok = np.array_equal(f[..., t], refs[t]) if var == M.VARIANT_EXACT else np.abs(f[..., t] - refs[t]).max() < 1e-12
            if not ok:
                bad += 1; print("run_uw batch MISMATCH", ncrms, nx, nz, T, t, var, np.abs(f[..., t] - refs[t]).max())
print("run_uw batch cases", n // 2, "bad", bad, flush=True)

M.set_variant(M.VARIANT_EXACT)
for it in range(n):
    ncrms, nx, nz = rshape()
    T = int(rng.integers(1, 4))
    base = O.make_inputs(ncrms, nx, nz, seed=300 + it, dist=3)
    fs = [O.make_inputs(ncrms, nx, nz, seed=31000 + 10 * it + t, dist=3)["f"] for t in range(T)]
    inp = dict(base)
    if T > 1:
        inp["f"] = np.asfortranarray(np.stack(fs, axis=-1)); inp["flux"] = np.asfortranarray(np.stack([base["flux"]] * T, axis=-1))
    else:
        inp["f"] = fs[0]
    d = {k: to_dev(v) for k, v in inp.items()}
    p = M.Plan(ncrms, nx, nz, T)
    p.import_device(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["adz"], d["flux"])
    fo = torch.empty_like(d["f"])
    p.export_device(fo)
    p.sync()
    if not torch.equal(fo, d["f"]):
        bad += 1; print("import/export round trip MISMATCH", ncrms, nx, nz, T)
    p.run(); p.export_device(fo); p.sync()
    q = M.Plan(ncrms, nx, nz, T)
    q.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"]); q.run(); q.sync()
    f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
    q.download(f, fl)
    p.close(); q.close()
    if not np.array_equal(to_host(fo), f):
        bad += 1; print("device import vs host upload MISMATCH", ncrms, nx, nz, T)
    for t in range(T):
        f_ref, _ = O.advect(dict(base, f=fs[t].copy()), nthreads=4)
        ft = f[..., t] if T > 1 else f
        if not np.array_equal(ft, f_ref):
            bad += 1; print("plan run MISMATCH", ncrms, nx, nz, T, t)
print("import / export / run cases", n, "bad", bad, flush=True)

for it in range(n // 2):
    ncrms, nx, nz = rshape(even=True)
    inp = O.make_inputs(ncrms, nx, nz, seed=800 + it, dist=int(rng.integers(1, 4)), dtype=np.float32)
    p = M.Plan(ncrms, nx, nz, 1, dtype=np.float32)
    d = {k: to_dev(v) for k, v in inp.items()}
    p.import_device(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["adz"], d["flux"])
    p.run()
    fo = torch.empty_like(d["f"]); p.export_device(fo); p.sync(); p.close()
    f_ref, _ = O.advect(inp, nthreads=4)
    if not np.array_equal(to_host(fo), f_ref):
        bad += 1; print("fp32 plan MISMATCH", ncrms, nx, nz, np.abs(to_host(fo).astype(np.float64) - f_ref).max())
print("fp32 plan cases", n // 2, "bad", bad, flush=True)
print("TOTAL bad", bad)
sys.exit(1 if bad else 0)
