#!/usr/bin/env python3
"""The shader clock a kernel actually runs at, measured INSIDE the kernel the way the guide asks
(/opt/skills/guides/MI355X_MICROARCH.md, power / clock section): every wave reads s_memtime (shader cycles) and
s_memrealtime (the constant 100-MHz counter) at its start and at its end; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz,
median over the waves of ONE launch that follows >= 2 s of back-to-back launches of the same kernel on live data
(cold field sets, cycled) -- so the figure is the clock the power management has settled on, not the ramp.

Needs a -DMPDWM_STAMPS build of the library (MPDATA_HIP_LIB); with -DMPDWM_ABL_FIRSTPASS / -DMPDWM_ABL_NOCOMPUTE
added, the same for the ablation builds (wrong results by design, timing only):
    bash tools/build_variants.sh "_stamps -DMPDWM_STAMPS" "_stfirst -DMPDWM_STAMPS -DMPDWM_ABL_FIRSTPASS" \
                                 "_stnocomp -DMPDWM_STAMPS -DMPDWM_ABL_NOCOMPUTE"
usage: python tools/inkernel_clock.py [--tracers 1|25] [--seconds 2.5] [--out file.json]   (tools/inkernel_clock.sh runs all)"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import codesign_kernels_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--ncrms", type=int, default=65536)
ap.add_argument("--nx", type=int, default=32)
ap.add_argument("--nz", type=int, default=28)
ap.add_argument("--tracers", type=int, default=1)
ap.add_argument("--sets", type=int, default=0)
ap.add_argument("--seconds", type=float, default=2.5)
ap.add_argument("--out", default="")
a = ap.parse_args()
ver = M.version()
assert "MPDWM_STAMPS" in ver, "needs a -DMPDWM_STAMPS build (MPDATA_HIP_LIB)"
M.set_variant(M.VARIANT_FAST)
dev = torch.device("cuda", 0)
ncrms, nx, nz, T = a.ncrms, a.nx, a.nz, a.tracers
nsets = a.sets or (10 if T == 1 else 3)
sh = M.shapes(ncrms, nx, nz, 1)
small = {k: torch.empty(sh[k], dtype=torch.float64, device=dev) for k in ("rho", "rhow", "adz", "flux")}
for k in small:
    M.fill_synthetic(small[k], k, 100, 1)
ftmp = torch.empty(sh["f"], dtype=torch.float64, device=dev)
u = torch.empty(sh["u"], dtype=torch.float64, device=dev)
w = torch.empty(sh["w"], dtype=torch.float64, device=dev)
plans = []
for s in range(nsets):
    M.fill_synthetic(u, "u", 100 + 31 * s, 1)
    M.fill_synthetic(w, "w", 100 + 31 * s, 1)
    p = M.Plan(ncrms, nx, nz, T)
    p.set_stream()
    p.set_timing(False)
    p.import_device(None, u, w, small["rho"], small["rhow"], small["adz"], None)
    for t in range(T):
        M.fill_synthetic(ftmp, "f", 100 + s * T + t, 1)
        p.import_device(ftmp, flux=small["flux"], first_tracer=t)
    plans.append(p)
torch.cuda.synchronize()
ntiles = (ncrms + 1) // 2
# stamp slots: 8 words per wave of the launch's grid (one tracer: a wave per tile; batches: the per-XCD tracer walk)
if T == 1:
    nwaves = (ntiles + 3) // 4 * 4
else:
    per_xcd = (ntiles + 7) // 8 * ((T + 1) // 2)
    nwaves = 8 * ((per_xcd + 3) // 4) * 4
dbg = torch.zeros(nwaves * 8, dtype=torch.int64, device=dev)
M.lib().mpdata_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
# >= `seconds` of back-to-back launches (the queue is kept fed: one synchronisation every 32 launches), the last one
# of them is the launch whose stamps are read: every launch overwrites the same slots
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < a.seconds:
    for _ in range(32 if T == 1 else 4):
        plans[n % nsets].run()
        n += 1
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
k = 16 if T == 1 else 3
e0.record()
for _ in range(k):
    plans[n % nsets].run()
    n += 1
e1.record()
torch.cuda.synchronize()
busy_s = time.perf_counter() - t0
ms = e0.elapsed_time(e1) / k
s = dbg.cpu().numpy().reshape(nwaves, 8)
M.lib().mpdata_set_debug_buffer(None)
for p in plans:
    p.close()
s = s[s[:, 1] > 0]          # waves that ran (grid padding writes nothing)
r0, r1, c0, c1 = (s[:, j].astype(np.float64) for j in range(4))
life_us = (r1 - r0) / 100.0
ok = life_us > 1.0
clk = (c1 - c0)[ok] / (life_us[ok] * 1e-6) / 1e9
span_us = (r1.max() - r0.min()) / 100.0
out = {"what": "in-kernel shader clock: d(s_memtime) / d(s_memrealtime) x 100 MHz per wave, one launch after "
               "%.1f s of back-to-back cold launches of the same kernel (%d launches)" % (busy_s, n),
       "build": ver.strip(), "shape": {"ncrms": ncrms, "nx": nx, "nz": nz, "tracers": T}, "field_sets": nsets,
       "waves_stamped": int(ok.sum()),
       "shader_clock_GHz": {"median": float(np.median(clk)), "p05": float(np.percentile(clk, 5)),
                            "p95": float(np.percentile(clk, 95)), "mean": float(clk.mean())},
       "wave_lifetime_us_median": float(np.median(life_us[ok])),
       "launch_span_us_first_start_to_last_end": float(span_us),
       "ms_per_plan_run_hip_events_last_%d" % k: ms}
print(json.dumps(out))
if a.out:
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
