#!/usr/bin/env python3
"""Time of the reference-layout device call (x-march kernel; nz > 64: see mpdata_core.hip), default ncrms=65536 nx=32
nz=28, 1 and 25 tracers.  usage: ref_bench.py [--ncrms N --nx X --nz Z --variant fast|exact --one]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import codesign_kernels_amd as M
ap = argparse.ArgumentParser()
ap.add_argument("--ncrms", type=int, default=65536)
ap.add_argument("--nx", type=int, default=32)
ap.add_argument("--nz", type=int, default=28)
ap.add_argument("--variant", default="fast")
ap.add_argument("--one", action="store_true", help="one tracer only")
A = ap.parse_args()
M.set_variant(M.VARIANT_FAST if A.variant == "fast" else M.VARIANT_EXACT)
dev = torch.device("cuda", 0)
ncrms, nx, nz = A.ncrms, A.nx, A.nz
for T, nb, steps in (((1, 16, 100),) if A.one else ((1, 16, 100), (25, 2, 6))):
    sh = M.shapes(ncrms, nx, nz, T)
    d = {k: M.empty_staggered(sh[k], k, torch.float64, dev) for k in ("u", "w", "rho", "rhow", "adz", "flux")}
    for k in d: M.fill_synthetic(d[k], k, 100, 1)
    fs = []
    for b in range(nb):
        f = M.empty_staggered(sh["f"], "f", torch.float64, dev)
        if T == 1: M.fill_synthetic(f, "f", 100 + b, 1)
        else:
            for t in range(T): M.fill_synthetic(f[t], "f", 100 + b * T + t, 1)
        fs.append(f)
    run = lambda i: M.advect_scalar2D(fs[i % nb], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"])
    for i in range(steps): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): run(i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    ab = M.algorithmic_bytes(ncrms, nx, nz, T)
    print(f"device call T={T}: {ms:.4f} ms  {ncrms*nx*(nz-1)*T/ms/1e6:.1f} Gcu/s  frac {ab/ms/1e6/8000:.3f}")
    del fs, d
    torch.cuda.empty_cache()
