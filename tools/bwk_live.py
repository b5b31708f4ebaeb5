#!/usr/bin/env python3
"""biharmonic_wk_scalar on LIVE data for seconds (power sampling, tools/power_side.sh) and as a timing:
the routine scales the field by ~1e-13 per call (rrearth twice), so a loop on one array computes on zeros after
25 calls.  Here: S arrays of 2 GB, launches cycle through them, every array is drawn afresh after 12 calls.
usage: python tools/bwk_live.py [--seconds T] [--sets S] [--decayed]   (--decayed: one array, never refreshed)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import codesign_kernels_amd.bwk as K
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=5.0)
ap.add_argument("--sets", type=int, default=8)
ap.add_argument("--decayed", action="store_true")
ap.add_argument("--variant", default="fast")
a = ap.parse_args()
K.set_variant(K.VARIANT_FAST if a.variant == "fast" else K.VARIANT_EXACT)
nelemd, nlev, qsize = 5400, 72, 40
g = torch.Generator(device="cuda").manual_seed(11)
S = 1 if a.decayed else a.sets
qs = [torch.rand((nelemd, qsize, nlev, 4, 4), dtype=torch.float64, device="cuda", generator=g) for _ in range(S)]
el = torch.rand((nelemd, 144), dtype=torch.float64, device="cuda", generator=g)
dv = torch.rand((4, 4), dtype=torch.float64, device="cuda", generator=g)
ab = K.algorithmic_bytes(nelemd, nlev, qsize)
for _ in range(40 if a.decayed else 3):
    for q in qs:
        K.biharmonic_wk_scalar(el, q, dv)
torch.cuda.synchronize()
t_end = time.time() + a.seconds
tot_ms, n = 0.0, 0
while time.time() < t_end:
    if not a.decayed:
        for q in qs:
            q.uniform_(0.0, 1.0, generator=g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        for q in qs:
            K.biharmonic_wk_scalar(el, q, dv)
    e1.record()
    torch.cuda.synchronize()
    tot_ms += e0.elapsed_time(e1); n += reps * S
ms = tot_ms / n
print("bwk %s %s: %.4f ms per call, %.2f TB/s = %.3f of 8 TB/s, finite %s" % (
    a.variant, "decayed field (zeros)" if a.decayed else f"live data ({S} arrays, redrawn every {reps} calls)",
    ms, ab / ms / 1e9, ab / ms / 1e9 / 8.0, bool(torch.isfinite(qs[0]).all())))
