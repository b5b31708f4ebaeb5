#!/usr/bin/env python3
"""Per-wave timeline of the headline kernel (needs the -DMPDWM_STAMPS build of the library:
bash tools/build_variants.sh "_stamps -DMPDWM_STAMPS"; MPDATA_HIP_LIB=.../libmpdata_hip_stamps.so).

Every wave of mpdata_advect_wm_kernel records its start / end on the 100-MHz real-time counter and on
the shader clock, the shader cycles it spent in its counted DMA waits, and where it ran.  From that:
how many waves are alive over the launch (ramp, steady state, drain), what the launch loses at both
ends against "every slot busy from the first start to the last end", wave lifetimes by dispatch
round, and the share of a wave's life spent waiting for its fetches.

COLD protocol as in bench.py: every launch on a plan of its own (own f, u, w), wake-up launches cycle
through the sets.  usage: python tools/wave_timeline.py [--ncrms N] [--launches K] [--out file.json]"""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import codesign_kernels_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--ncrms", type=int, default=65536)
ap.add_argument("--nx", type=int, default=32)
ap.add_argument("--nz", type=int, default=28)
ap.add_argument("--sets", type=int, default=10)
ap.add_argument("--launches", type=int, default=8)
ap.add_argument("--out", default="gpurun_out/wave_timeline.json")
ap.add_argument("--raw", default="", help="also save the raw stamps of the first measured launch (.npy, [wave][8])")
a = ap.parse_args()
assert "MPDWM_STAMPS" in M.version(), "needs the -DMPDWM_STAMPS build (MPDATA_HIP_LIB)"
M.set_variant(M.VARIANT_FAST)
dev = torch.device("cuda", 0)
ncrms, nx, nz = a.ncrms, a.nx, a.nz
sh = M.shapes(ncrms, nx, nz, 1)
small = {k: torch.empty(sh[k], dtype=torch.float64, device=dev) for k in ("rho", "rhow", "adz", "flux")}
for k in small:
    M.fill_synthetic(small[k], k, 100, 1)
ftmp = torch.empty(sh["f"], dtype=torch.float64, device=dev)
u = torch.empty(sh["u"], dtype=torch.float64, device=dev)
w = torch.empty(sh["w"], dtype=torch.float64, device=dev)
plans = []
for s in range(a.sets):
    M.fill_synthetic(u, "u", 100 + 31 * s, 1)
    M.fill_synthetic(w, "w", 100 + 31 * s, 1)
    M.fill_synthetic(ftmp, "f", 100 + s, 1)
    p = M.Plan(ncrms, nx, nz, 1)
    p.set_stream()
    p.set_timing(False)
    p.import_device(ftmp, u, w, small["rho"], small["rhow"], small["adz"], small["flux"])
    plans.append(p)
torch.cuda.synchronize()
ntiles = (ncrms + 1) // 2
nwaves = (ntiles + 7) // 8 * 8
dbg = torch.zeros(nwaves * 8, dtype=torch.int64, device=dev)
# wake-up: 150 cold launches without stamps being read (the buffer is written all the same)
M.lib().mpdata_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
for i in range(150):
    plans[i % a.sets].run()
torch.cuda.synchronize()

recs = []
for it in range(a.launches):
    dbg.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # two un-stamped-for-us launches in front so that the measured one starts on a busy, clocked-up chip
    plans[(3 * it) % a.sets].run()
    plans[(3 * it + 1) % a.sets].run()
    torch.cuda.synchronize()
    dbg.zero_()
    torch.cuda.synchronize()
    e0.record()
    plans[(3 * it + 2) % a.sets].run()
    e1.record()
    torch.cuda.synchronize()
    s = dbg.cpu().numpy().reshape(nwaves, 8)[:ntiles]
    if a.raw and it == 0:
        np.save(a.raw, s)
    r0, r1, c0, c1, wt, hw, xcc = (s[:, j].astype(np.float64) for j in range(7))
    ok = r1 > 0
    assert ok.all(), f"{(~ok).sum()} waves wrote no stamps"
    t0 = r0.min()
    ts, te = (r0 - t0) / 100.0, (r1 - t0) / 100.0          # microseconds
    span = te.max()
    life = te - ts
    clk = (c1 - c0) / (life * 1e-6) / 1e9
    # waves alive on a 0.5-us grid
    grid = np.arange(0.0, span + 0.5, 0.5)
    alive = (np.searchsorted(np.sort(ts), grid, side="right") - np.searchsorted(np.sort(te), grid, side="right")).astype(float)
    peak = alive.max()
    lost = (peak - alive) * 0.5 / peak                       # idle slot-time as microseconds of the whole chip
    half = len(grid) // 2
    order = np.argsort(ts)
    rounds = [float(life[order[i:i + int(peak)]].mean()) for i in range(0, ntiles, int(peak))]
    ends_last = np.sort(te)[-int(peak):]
    rec = {"event_ms": e0.elapsed_time(e1), "span_us_first_start_to_last_end": float(span),
           "waves": int(ntiles), "peak_waves_alive": int(peak),
           "lifetime_us": {"median": float(np.median(life)), "p05": float(np.percentile(life, 5)),
                           "p95": float(np.percentile(life, 95)), "by_dispatch_round_mean": rounds},
           "sum_of_lifetimes_over_span_times_peak": float(life.sum() / (span * peak)),
           "idle_equiv_us": {"first_half": float(lost[:half].sum()), "second_half": float(lost[half:].sum())},
           "time_from_last_wave_start_to_end_us": float(span - ts.max()),
           "last_round_end_spread_us": {"p05_to_max": float(ends_last.max() - np.percentile(ends_last, 5)),
                                        "median_to_max": float(ends_last.max() - np.median(ends_last))},
           "start_spread_first_round_us": float(np.sort(ts)[int(peak) - 1]),
           "dma_wait_share_of_wave_cycles": {"median": float(np.median(wt / (c1 - c0))),
                                             "first_round": float(np.median((wt / (c1 - c0))[order[:int(peak)]])),
                                             "last_round": float(np.median((wt / (c1 - c0))[order[-int(peak):]]))},
           "shader_clock_GHz": {"median": float(np.median(clk)), "p05": float(np.percentile(clk, 5))},
           "alive_every_10us": [int(x) for x in alive[::20]]}
    recs.append(rec)
M.lib().mpdata_set_debug_buffer(None)
for p in plans:
    p.close()
med = lambda key: float(np.median([r[key] for r in recs]))
out = {"what": "per-wave stamps of mpdata_advect_wm_kernel<double,32,4,true> (FAST, -DMPDWM_STAMPS build), cold launches",
       "shape": {"ncrms": ncrms, "nx": nx, "nz": nz}, "launches": recs,
       "median": {"event_ms": med("event_ms"), "span_us": med("span_us_first_start_to_last_end"),
                  "idle_equiv_us_first_half": float(np.median([r["idle_equiv_us"]["first_half"] for r in recs])),
                  "idle_equiv_us_second_half": float(np.median([r["idle_equiv_us"]["second_half"] for r in recs])),
                  "lifetime_us": float(np.median([r["lifetime_us"]["median"] for r in recs])),
                  "dma_wait_share": float(np.median([r["dma_wait_share_of_wave_cycles"]["median"] for r in recs]))}}
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
with open(a.out, "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out["median"]))
for r in recs[:3]:
    print("event %.4f ms span %.1f us peak %d alive/10us %s" % (r["event_ms"], r["span_us_first_start_to_last_end"],
                                                                 r["peak_waves_alive"], r["alive_every_10us"]))
    print("   lifetime by round", ["%.1f" % x for x in r["lifetime_us"]["by_dispatch_round_mean"]],
          "idle-equivalent us first/second half %.1f / %.1f" % (r["idle_equiv_us"]["first_half"], r["idle_equiv_us"]["second_half"]),
          "wait share %.2f" % r["dma_wait_share_of_wave_cycles"]["median"])
