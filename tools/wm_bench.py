#!/usr/bin/env python3
"""Kernel time of the plan API (wave-major layout) against the device-pointer call (reference
layout, x-march kernel) at ncrms=65536 nx=32 nz=28: 1 tracer (each launch on its own buffer) and
25 tracers.  usage: python tools/wm_bench.py [--variant fast|exact] [--steps N] [--no-t25]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import codesign_kernels_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="fast")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--nbuf", type=int, default=16)
ap.add_argument("--ncrms", type=int, default=65536)
ap.add_argument("--no-t25", action="store_true")
ap.add_argument("--no-ref", action="store_true")
ap.add_argument("--no-t1", action="store_true")
ap.add_argument("--t25-steps", type=int, default=6)
ap.add_argument("--tracers", type=int, default=25, help="tracers of the batch run")
a = ap.parse_args()
M.set_variant(M.VARIANT_FAST if a.variant == "fast" else M.VARIANT_EXACT)
dev = torch.device("cuda", 0)
ncrms, nx, nz = a.ncrms, 32, 28
cells = ncrms * nx * (nz - 1)

def problem(ntr):
    sh = M.shapes(ncrms, nx, nz, 1)
    d = {k: M.empty_staggered(sh[k], k, torch.float64, dev) for k in ("u", "w", "rho", "rhow", "adz", "flux", "f")}
    for k in d:
        M.fill_synthetic(d[k], k, 100, 1)
    return d

def timed(fn, steps, warm):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps

d = problem(1)
# ---- plan, 1 tracer per launch, nbuf field buffers
if a.no_t1:
    a.nbuf, a.steps = 1, 1
p = M.Plan(ncrms, nx, nz, a.nbuf)
p.set_stream()
p.import_device(None, d["u"], d["w"], d["rho"], d["rhow"], d["adz"], None)
for t in range(a.nbuf):
    M.fill_synthetic(d["f"], "f", 100 + t, 1)
    p.import_device(d["f"], flux=d["flux"], first_tracer=t)
torch.cuda.synchronize()
ms = timed(lambda i: p.run(i % a.nbuf, 1), a.steps, a.steps)
ab = M.algorithmic_bytes(ncrms, nx, nz, 1)
print(f"plan wave-major  T=1 : {ms:.4f} ms  {cells / ms / 1e6:.1f} Gcu/s  {ab / ms / 1e6:.0f} GB/s  frac {ab / ms / 1e6 / 8000:.3f}")
p.close()
if not a.no_ref:
    fs = []
    for t in range(a.nbuf):
        f = M.empty_staggered(d["f"].shape, "f", torch.float64, dev)
        M.fill_synthetic(f, "f", 100 + t, 1)
        fs.append(f)
    ms = timed(lambda i: M.advect_scalar2D(fs[i % a.nbuf], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"]), a.steps, a.steps)
    print(f"device call x-march T=1 : {ms:.4f} ms  {cells / ms / 1e6:.1f} Gcu/s  frac {ab / ms / 1e6 / 8000:.3f}")
    del fs
if not a.no_t25:
    T = a.tracers
    p = M.Plan(ncrms, nx, nz, T)
    p.set_stream()
    p.import_device(None, d["u"], d["w"], d["rho"], d["rhow"], d["adz"], None)
    for t in range(T):
        M.fill_synthetic(d["f"], "f", 100 + t, 1)
        p.import_device(d["f"], flux=d["flux"], first_tracer=t)
    torch.cuda.synchronize()
    ms = timed(lambda i: p.run(), a.t25_steps, 3 if a.t25_steps > 1 else 0)
    ab = M.algorithmic_bytes(ncrms, nx, nz, T)
    print(f"plan wave-major  T={T}: {ms:.4f} ms  {cells * T / ms / 1e6:.1f} Gcu/s  {ab / ms / 1e6:.0f} GB/s  frac {ab / ms / 1e6 / 8000:.3f}")
    p.close()
