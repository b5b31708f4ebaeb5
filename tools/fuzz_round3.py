#!/usr/bin/env python3
"""One-off wider sweep than the test suite (round 3's new kernels), GPU:
  * mpdata_plan_run_uw, EXACT, bitwise vs the oracle: N random shapes (even and odd ncrms, nz 3..40, nx 1..60)
  * one tracer, FAST, plan run: max |df| < 1e-12 and flux within 1e-12 (relative to max |flux|) on conditioned inputs
  * tracer batches (2..7 tracers) FAST vs the oracle (max |df| < 1e-12 on conditioned inputs) and EXACT bitwise
  * nlk: random meshes, ragged, every kernel form, EXACT bitwise
usage: python tools/fuzz_round3.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import codesign_kernels_amd as M
import codesign_kernels_amd.nlk as K
from oracle import oracle as O
from oracle import nlk as N
from util import to_dev, to_host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
O.build_lib(); N.build_lib()
rng = np.random.default_rng(777)
bad = 0
for it in range(n):
    ncrms = int(rng.integers(1, 400)); nx = int(rng.integers(1, 61)); nz = int(rng.integers(3, 41))
    dist = int(rng.integers(1, 4))
    M.set_variant(M.VARIANT_EXACT)
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
print("run_uw cases", n, "bad", bad)
# one tracer, FAST (the headline kernel's form: 7-operation extrema, ring sums, merged U / dW ring value): f and flux
M.set_variant(M.VARIANT_FAST)
for it in range(n):
    ncrms = int(rng.integers(1, 400)); nx = int(rng.integers(1, 61)); nz = int(rng.integers(3, 65))
    inp = O.make_inputs(ncrms, nx, nz, seed=40000 + it, dist=1)
    p = M.Plan(ncrms, nx, nz, 1)
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"]); p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
    p.download(f, fl); p.close()
    f_ref, fl_ref = O.advect(inp, nthreads=4)
    ef = np.abs(f - f_ref).max(); efl = np.abs(fl - fl_ref).max() / max(np.abs(fl_ref).max(), 1e-300)
    if not (ef < 1e-12 and efl < 1e-12):
        bad += 1; print("fast T=1 MISMATCH", ncrms, nx, nz, ef, efl)
print("fast one-tracer cases", n, "bad", bad)
for it in range(n // 2):
    ncrms = int(rng.integers(1, 300)); nx = int(rng.integers(1, 50)); nz = int(rng.integers(3, 65)); T = int(rng.integers(2, 8))
    base = O.make_inputs(ncrms, nx, nz, seed=100 + it, dist=1)
    fs = [O.make_inputs(ncrms, nx, nz, seed=20000 + 10 * it + t, dist=1)["f"] for t in range(T)]
    inp = dict(base); inp["f"] = np.asfortranarray(np.stack(fs, axis=-1)); inp["flux"] = np.asfortranarray(np.stack([base["flux"]] * T, axis=-1))
    for var in (M.VARIANT_EXACT, M.VARIANT_FAST):
        M.set_variant(var)
        p = M.Plan(ncrms, nx, nz, T)
        p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"]); p.run(); p.sync()
        f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
        p.download(f, fl); p.close()
        for t in range(T):
            f_ref, _ = O.advect(dict(base, f=fs[t].copy()), nthreads=4)
            ok = np.array_equal(f[..., t], f_ref) if var == M.VARIANT_EXACT else np.abs(f[..., t] - f_ref).max() < 1e-12
            if not ok:
                bad += 1; print("batch MISMATCH", ncrms, nx, nz, T, t, var, np.abs(f[..., t] - f_ref).max())
print("batch cases", n // 2, "bad", bad)
K.set_variant(K.VARIANT_EXACT)
for it in range(n // 2):
    nE = int(rng.integers(1, 3000)); nC = int(rng.integers(1, 500)); nV = int(rng.integers(1, 140)); nA = int(rng.integers(1, 14))
    nvldim = nV + int(rng.integers(0, 5))
    inp = N.make_inputs(nE, nC, nV, nA, seed=300 + it, nvldim=nvldim, ragged=True)
    ref = N.high_order_flux(inp)
    for mode in (-1, 0, 1, 2):
        K.set_kernel(mode)
        d = {}
        for k in K.INT_KEYS:
            d[k] = torch.from_numpy(np.ascontiguousarray(np.asarray(inp[k], dtype=np.int32).T)).to("cuda:0")
        for k in K.REAL_KEYS:
            d[k] = torch.from_numpy(np.ascontiguousarray(np.asarray(inp[k], dtype=np.float64).T)).to("cuda:0")
        out = torch.full((nE, nvldim), -3.5, dtype=torch.float64, device="cuda:0")
        K.high_order_flux(d, inp["nVertLevels"], inp["coef3rdOrder"], out)
        torch.cuda.synchronize()
        o = np.asfortranarray(out.cpu().numpy().T)
        if not (np.array_equal(o[:nV], ref[:nV]) and np.all(o[nV:] == -3.5)):
            bad += 1; print("nlk MISMATCH", nE, nC, nV, nA, nvldim, mode, np.abs(o[:nV] - ref[:nV]).max())
K.set_kernel(-1)
print("nlk cases", n // 2, "bad", bad)
sys.exit(1 if bad else 0)
