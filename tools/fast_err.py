#!/usr/bin/env python3
"""Error of the FAST variant (plan API) against the oracle: max |df| on the conditioned law, relative L1
on the reference-raw law (tolerances: 1e-12 / 1e-14).  usage: python tools/fast_err.py [ncrms]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import codesign_kernels_amd as M
from oracle import oracle as O
ncrms = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M.set_variant(M.VARIANT_FAST)
for dist, name in ((1, "conditioned"), (2, "reference-raw"), (3, "raw-signed")):
    inp = O.make_inputs(ncrms, 32, 28, seed=100, dist=dist)
    f_ref, fl_ref = O.advect(inp, nthreads=8)
    p = M.Plan(ncrms, 32, 28)
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
    p.download(f, fl); p.close()
    print(f"{name:14s} max|df| {np.abs(f - f_ref).max():.3e}  max|dflux| {np.abs(fl - fl_ref)[:, :-1].max():.3e}  "
          f"rel-L1 f {O.rel_l1(f, f_ref):.3e}  rel-L1 flux {O.rel_l1(fl[:, :-1], fl_ref[:, :-1]):.3e}  max|f| {np.abs(f_ref).max():.3g}")
