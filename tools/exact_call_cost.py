#!/usr/bin/env python3
"""Wall time per reference-layout device call (mpdata_advect_scalar2d_device) in a back-to-back loop, EXACT with its park
array in stream order around the call against MPDATA_EXACT_FLUX=sum (no park) and FAST: what the stream-ordered
allocation of the park array costs a caller that loops.  usage: python tools/exact_call_cost.py"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for label, env in (("EXACT, flux parked", {}), ("EXACT, MPDATA_EXACT_FLUX=sum", {"MPDATA_EXACT_FLUX": "sum"}),
                       ("FAST", {"MPDATA_VARIANT": "fast"})):
        subprocess.run([sys.executable, __file__, label], env=dict(os.environ, **env), check=True)
    sys.exit(0)
import torch
import codesign_kernels_amd as M
M.set_variant(M.VARIANT_FAST if os.environ.get("MPDATA_VARIANT") == "fast" else M.VARIANT_EXACT)
ncrms, nx, nz = 65536, 32, 28
sh = M.shapes(ncrms, nx, nz, 1)
d = {k: torch.empty(sh[k], dtype=torch.float64, device="cuda") for k in ("f", "u", "w", "rho", "rhow", "adz", "flux")}
for k in d:
    M.fill_synthetic(d[k], k, 100, 1)
fs = [d["f"].clone() for _ in range(8)]
for i in range(16):
    M.advect_scalar2D(fs[i % 8], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"])
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for i in range(n):
    M.advect_scalar2D(fs[i % 8], d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"])
torch.cuda.synchronize()
print("%-32s %.4f ms per call (wall, %d calls back to back)" % (sys.argv[1], (time.perf_counter() - t0) / n * 1e3, n))
