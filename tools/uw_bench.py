#!/usr/bin/env python3
"""Cold kernel times at ncrms=65536 nx=32 nz=28 (FAST): every launch on a field set of its own.
  plan T=1   : mpdata_plan_run on a plan per set (own f, u, w)            -- the headline protocol
  run_uw T=1 : mpdata_plan_run_uw, u / w reference-layout device arrays, distinct per step
  import u+w : the fused layout conversion of u and w alone; import f / export f
usage: python tools/uw_bench.py [--steps N] [--sets N] [--no-plan] [--no-uw] [--no-conv]
MPDATA_HIP_LIB selects an experiment build of the library."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import codesign_kernels_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="fast")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--sets", type=int, default=12)
ap.add_argument("--ncrms", type=int, default=65536)
ap.add_argument("--nx", type=int, default=32)
ap.add_argument("--nz", type=int, default=28)
ap.add_argument("--no-plan", action="store_true")
ap.add_argument("--no-uw", action="store_true")
ap.add_argument("--no-conv", action="store_true")
ap.add_argument("--batch", type=int, default=0, help="also: run_uw on plans of this many tracers (fresh u, w per step)")
ap.add_argument("--batch-steps", type=int, default=12, help="timed steps of the tracer-batch loops")
a = ap.parse_args()
M.set_variant(M.VARIANT_FAST if a.variant == "fast" else M.VARIANT_EXACT)
dev = torch.device("cuda", 0)
ncrms, nx, nz = a.ncrms, a.nx, a.nz
cells = ncrms * nx * (nz - 1)
ab = M.algorithmic_bytes(ncrms, nx, nz, 1)
sh = M.shapes(ncrms, nx, nz, 1)


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


small = {k: torch.empty(sh[k], dtype=torch.float64, device=dev) for k in ("rho", "rhow", "adz", "flux")}
for k in small:
    M.fill_synthetic(small[k], k, 100, 1)
ftmp = torch.empty(sh["f"], dtype=torch.float64, device=dev)
sets = []
for s in range(a.sets):
    u = M.empty_staggered(sh["u"], "u", torch.float64, dev)
    w = M.empty_staggered(sh["w"], "w", torch.float64, dev)
    M.fill_synthetic(u, "u", 100 + 31 * s, 1)
    M.fill_synthetic(w, "w", 100 + 31 * s, 1)
    M.fill_synthetic(ftmp, "f", 100 + s, 1)
    p = M.Plan(ncrms, nx, nz, 1)
    p.set_stream()
    p.set_timing(False)
    p.import_device(ftmp, u, w, small["rho"], small["rhow"], small["adz"], small["flux"])
    sets.append((p, u, w))
torch.cuda.synchronize()
n = a.sets
if not a.no_plan:
    ms = timed(lambda i: sets[i % n][0].run(), a.steps, a.steps)
    print(f"plan   T=1 cold : {ms:.4f} ms  {cells / ms / 1e6:.1f} Gcu/s  frac {ab / ms / 1e6 / 8000:.4f}")
if not a.no_uw:
    ms = timed(lambda i: sets[i % n][0].run_uw(sets[i % n][1], sets[i % n][2]), a.steps, a.steps)
    print(f"run_uw T=1 cold : {ms:.4f} ms  {cells / ms / 1e6:.1f} Gcu/s  frac {ab / ms / 1e6 / 8000:.4f}")
if not a.no_conv:
    nb = ftmp.numel() * 8
    ms = timed(lambda i: sets[i % n][0].import_device(None, sets[i % n][1], sets[i % n][2]), 20, 5)
    print(f"import u+w      : {ms:.4f} ms  {2 * (sets[0][1].numel() + sets[0][2].numel()) * 8 / ms / 1e6:.0f} GB/s (read + write)")
    ms = timed(lambda i: sets[i % n][0].import_device(ftmp), 20, 5)
    print(f"import f        : {ms:.4f} ms  {2 * nb / ms / 1e6:.0f} GB/s")
    ms = timed(lambda i: sets[i % n][0].export_device(ftmp), 20, 5)
    print(f"export f        : {ms:.4f} ms  {2 * nb / ms / 1e6:.0f} GB/s")
for p, _, _ in sets:
    p.close()
if a.batch > 1:   # tracer batches on fresh reference-layout u, w: MPDATA_RUN_UW=import selects the conversion path
    T = a.batch
    nb = max(3, min(8, int(torch.cuda.mem_get_info()[0] * 0.5 // (ftmp.numel() * 8 * T))))
    plans = []
    for s_ in range(nb):
        p = M.Plan(ncrms, nx, nz, T)
        p.set_stream(); p.set_timing(False)
        p.import_device(None, sets[s_ % n][1], sets[s_ % n][2], small["rho"], small["rhow"], small["adz"], None)
        for t in range(T):
            M.fill_synthetic(ftmp, "f", 300 + s_ * T + t, 1)
            p.import_device(ftmp, flux=small["flux"], first_tracer=t)
        plans.append(p)
    torch.cuda.synchronize()
    abT = M.algorithmic_bytes(ncrms, nx, nz, T)
    # (run() first: run_uw leaves no velocities in a plan)
    ms = timed(lambda i: plans[i % nb].run(), a.batch_steps, 4)
    print(f"plan   T={T} cold : {ms:.4f} ms  {cells * T / ms / 1e6:.1f} Gcu/s  frac {abT / ms / 1e6 / 8000:.4f}")
    ms = timed(lambda i: plans[i % nb].run_uw(sets[(i + 1) % n][1], sets[(i + 1) % n][2]), min(a.batch_steps, 12), 4)
    print(f"run_uw T={T} cold : {ms:.4f} ms  {cells * T / ms / 1e6:.1f} Gcu/s  frac {abT / ms / 1e6 / 8000:.4f}  ({os.environ.get('MPDATA_RUN_UW', 'direct')})")
    for p in plans:
        p.close()
