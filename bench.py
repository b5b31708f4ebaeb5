#!/usr/bin/env python3
"""bench.py -- MPDATA tracer-advection throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one call of advect_scalar2D (the fused HIP kernel) over one batch
of synthetic input that is already resident in HBM: BASELINE.json configs[2]
per GPU -- ncrms=65536, nx=32, nz=28, fp64, 1 tracer.  The ncrms axis is
embarrassingly parallel, so N GPUs run N shards of a global ncrms=N*65536
problem with no data-path collective (weak scaling); RCCL is only used for the
barrier / max-over-ranks of the timing (and, outside the timed region, by the
optional --scatter check).  Every step works on its own pristine copy of f
(the routine updates f in place), so no restore sits inside the timed region.

Prints ONE JSON line (rank 0).  `value` = cell-updates/s of the whole job;
`roofline` = algorithmic HBM bytes per launch / mean kernel time (HIP events
on the launch stream) against the 8 TB/s HBM3E peak; `cpu_baseline` = the
reference Fortran executable itself (oracle/_ref, built from /root/reference
by oracle/build_ref.py) timed on this box's host, 1 core, ncrms=4096.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 100 + 100 launches of 0.5 ms.  The first ~30 ms of GPU activity after idle run
    # 5-10 % slow (clock / power-state ramp, tools/step_series.py), so the warm-up is sized to
    # leave that transient outside the timed region
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed GPU wake-up before the warm-up steps: scratch launches of the same kernel for "
                         "about this long, so that the clock / power-state ramp after idle (DESIGN.md 4.4) is "
                         "over even when --warmup is small; 0 disables")
    ap.add_argument("--ncrms-per-gpu", type=int, default=65536)
    ap.add_argument("--nx", type=int, default=32)
    ap.add_argument("--nz", type=int, default=28)
    ap.add_argument("--tracers", type=int, default=1)
    ap.add_argument("--variant", choices=["exact", "fast"], default=os.environ.get("MPDATA_VARIANT", "fast"),
                    help="fast: FMA contraction (max|df| < 1e-12 vs the reference, tests/test_hip_parity.py); "
                         "exact: bit-identical f")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                    help="f64: the headline (BASELINE.json fp64); f32: the reference's precision switch")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 side measurement")
    ap.add_argument("--no-bwk", action="store_true", help="skip the biharmonic_wk_scalar side measurement")
    ap.add_argument("--aligned", action="store_true",
                    help="allocate f, u, w with equally aligned bases (plain torch.empty) instead of the "
                         "staggered placement (DESIGN.md 4.4: 8 %% slower at ncrms=65536)")
    ap.add_argument("--tile", type=int, default=-1)
    ap.add_argument("--dist", type=int, default=1, help="1 conditioned, 2 reference-raw, 3 raw-signed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the 25-tracer side measurement")
    ap.add_argument("--batched-tracers", type=int, default=25)
    ap.add_argument("--scatter", action="store_true",
                    help="also time the RCCL scatter/gather of a small problem (outside the timed region)")
    return ap.parse_args()


def cpu_baseline(nx, nz):
    """The reference program itself on the host (1 core; the reference is
    serial, mmf-mpdata-tracer/Makefile has no OpenMP).  Bounded sample:
    ncrms=4096, repeated runs, ~10-20 s."""
    from oracle import oracle as O
    ncrms = 4096
    cells = ncrms * nx * (nz - 1)
    inp = O.make_inputs(ncrms, nx, nz, seed=100, dist=O.DIST_CONDITIONED)
    out = {}
    if O.ref_exe(ncrms, nx, nz) is not None:
        times = []
        t_end = time.time() + 15.0
        while len(times) < 12 and (time.time() < t_end or len(times) < 3):
            _, _, t = O.run_reference(inp, want_outputs=False)
            times.append(t)
        t = statistics.median(times)
        out = {"value": cells / t, "unit": "cell-updates/s", "cores": 1, "kind": "reference",
               "sample": f"reference executable (amdflang -O3 -ffp-contract=off), ncrms={ncrms} nx={nx} "
                         f"nz={nz}, median of {len(times)} cold single calls (its own 'CPU Timing' line)",
               "seconds_per_call": t}
    # the C restatement, serial and on all host cores (this build's own OpenMP)
    O.build_lib()
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.advect(inp, nthreads=1)
        reps.append(time.perf_counter() - t0)
    port1 = cells / min(reps)
    nthr = O.max_threads()
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.advect(inp, nthreads=nthr)
        reps.append(time.perf_counter() - t0)
    portn = cells / min(reps)
    if not out:
        out = {"value": port1, "unit": "cell-updates/s", "cores": 1, "kind": "port",
               "sample": f"C restatement (gcc -O3 -ffp-contract=off), ncrms={ncrms}, best of 3"}
    out["port_serial"] = port1
    out["port_openmp"] = {"value": portn, "cores": nthr}
    return out


def copy_ceiling(torch, dev):
    """Measured device-to-device copy rate of this box ('also record a measured
    device-copy ceiling'): torch copy of a 538-MB buffer (the size of f), read + write bytes."""
    n = 65536 * 38 * 27
    a = torch.empty(n, dtype=torch.float64, device=dev).fill_(1.0)
    b = torch.empty_like(a)
    for _ in range(10):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2 * n * 8 / (e0.elapsed_time(e1) / 30 * 1e-3) / 1e9


def bench_bwk(torch, dev, steps, warmup, with_cpu):
    """Side measurement of the second kernel: biharmonic_wk_scalar on a
    cubed-sphere ne=30 mesh (5400 elements x 72 levels x 40 tracers, 2 GB of qtens), FAST
    variant, in place on resident data (random values; the arithmetic is data-independent)."""
    import codesign_kernels_amd.bwk as K
    nelemd, nlev, qsize = 5400, 72, 40
    K.set_variant(K.VARIANT_FAST)
    g = torch.Generator(device=dev).manual_seed(11)
    q = torch.rand((nelemd, qsize, nlev, 4, 4), dtype=torch.float64, device=dev, generator=g)
    el = torch.rand((nelemd, 144), dtype=torch.float64, device=dev, generator=g)
    dv = torch.rand((4, 4), dtype=torch.float64, device=dev, generator=g)
    for _ in range(warmup):
        K.biharmonic_wk_scalar(el, q, dv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        K.biharmonic_wk_scalar(el, q, dv)   # rrearth ~ 1e-7 twice per call: the field decays, stays finite
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = e0.elapsed_time(e1) / steps
    ab = K.algorithmic_bytes(nelemd, nlev, qsize)
    slabs = nelemd * nlev * qsize
    out = {"workload": f"atmosphere/biharmonic_wk_kernel.F90 biharmonic_wk_scalar: nelemd={nelemd} nlev={nlev} "
                       f"qsize={qsize} fp64, device-resident, in place",
           "value": slabs * steps / dt, "unit": "4x4-slab Laplacians/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
           "roofline": {"bound": "hbm", "achieved": ab / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab,
                        "kernel_ms_avg": kms}}
    if with_cpu:
        from oracle import bwk as B
        B.build_lib()
        cpu = {}
        if B.ref_exe(16) is not None:
            ts = sorted(B.run_reference(16)[2] for _ in range(5))
            cpu = {"value": 16 * nlev * qsize / ts[len(ts) // 2], "unit": "4x4-slab Laplacians/s", "cores": 1,
                   "kind": "reference", "sample": "reference executable (amdflang -O3 -ffp-contract=off), its shipped "
                   "size nelemd=16, median of 5 runs (its own 'CPU time' line)"}
        inp = B.make_inputs(64)
        t0 = time.perf_counter(); B.biharmonic(inp); t1 = time.perf_counter() - t0
        if not cpu:
            cpu = {"value": 64 * nlev * qsize / t1, "unit": "4x4-slab Laplacians/s", "cores": 1, "kind": "port",
                   "sample": "C restatement, nelemd=64"}
        cpu["port_serial"] = 64 * nlev * qsize / t1
        out["cpu_baseline"] = cpu
    return out


def bench_nlk(torch, dev, steps, warmup, with_cpu):
    """Side measurement of the third kernel: the MPAS-Ocean high-order flux loop
    nest at the reference's namelist size (nested_loops/nested.nml: 25600 edges, 2800 cells,
    100 levels, 10 cells per edge), FAST variant, device-resident."""
    import numpy as np
    import codesign_kernels_amd.nlk as K
    nE, nC, nV, nA = 25600, 2800, 100, 10
    K.set_variant(K.VARIANT_FAST)
    g = torch.Generator(device=dev).manual_seed(3)
    rnd = lambda *shape: torch.rand(shape, dtype=torch.float64, device=dev, generator=g)
    d = {"nAdvCellsForEdge": torch.full((nE,), nA, dtype=torch.int32, device=dev),
         "advCellsForEdge": torch.randint(1, nC + 1, (nE, nA), dtype=torch.int32, device=dev, generator=g),
         "minLevelCell": torch.ones((nC,), dtype=torch.int32, device=dev),
         "maxLevelCell": torch.clamp((rnd(nC) * nV * 2).round().to(torch.int32), 3, nV),
         "tracerCur": 15.0 * rnd(nC, nV), "normalThicknessFlux": 15.0 * (0.5 - rnd(nE, nV)),
         "advMaskHighOrder": torch.ones((nE, nV), dtype=torch.float64, device=dev),
         "advCoefs": 20.0 * rnd(nE, nA), "advCoefs3rd": 21.0 * rnd(nE, nA)}
    out = torch.zeros((nE, nV), dtype=torch.float64, device=dev)
    coef = float(np.float32(2.14))
    for _ in range(warmup):
        K.high_order_flux(d, nV, coef, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        K.high_order_flux(d, nV, coef, out)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = e0.elapsed_time(e1) / steps
    ab = K.algorithmic_bytes(nE, nC, nV, nV, nA)
    res = {"workload": f"nested_loops/nested.F90 high-order flux loop nest: nEdges={nE} nCells={nC} "
                       f"nVertLevels={nV} nAdv={nA} fp64, device-resident",
           "value": nE * nV * steps / dt, "unit": "edge-level fluxes/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
           "roofline": {"bound": "hbm", "achieved": ab / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab,
                        "kernel_ms_avg": kms,
                        "note": "62 MB per call: the working set fits the 256-MB Infinity Cache and the call is "
                                "short; launch-latency territory, not an HBM-bound measurement"}}
    if with_cpu:
        from oracle import nlk as N
        N.build_lib()
        inp = N.make_inputs(nE, nC, nV, nA, seed=1, ragged=False)
        best = min(_timeit(lambda: N.high_order_flux(inp)) for _ in range(3))
        res["cpu_baseline"] = {"value": nE * nV / best, "unit": "edge-level fluxes/s", "cores": 1, "kind": "port",
                               "sample": "C restatement of nested.F90:123-157, the namelist size, best of 3"}
    return res


def _timeit(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def make_problem(M, torch, dev, ncrms_loc, ncrms_glob, sl0, nx, nz, ntr, nbuf, dist, dtype=None):
    dtype = torch.float64 if dtype is None else dtype
    sh = M.shapes(ncrms_loc, nx, nz, ntr)
    # arrays placed as INTEGRATION.md advises a caller to: f, u, w at different offsets modulo
    # 1 KiB (HBM channel interleave); --aligned reproduces equally aligned bases instead
    alloc = (lambda shape, k: torch.empty(shape, dtype=dtype, device=dev)) if ALIGNED else \
            (lambda shape, k: M.empty_staggered(shape, k, dtype, dev))
    d = {k: alloc(sh[k], k) for k in ("u", "w", "rho", "rhow", "adz", "flux")}
    for k in d:
        M.fill_synthetic(d[k], k, 100, dist, ncrms_global=ncrms_glob, sl0=sl0)
    fs = []
    for b in range(nbuf):
        f = alloc(sh["f"], "f")
        # per-tracer / per-buffer seeds: distinct data, same law
        if ntr == 1:
            M.fill_synthetic(f, "f", 100 + b, dist, ncrms_global=ncrms_glob, sl0=sl0)
        else:
            for t in range(ntr):
                M.fill_synthetic(f[t], "f", 100 + b * ntr + t, dist, ncrms_global=ncrms_glob, sl0=sl0)
        fs.append(f)
    return d, fs


ALIGNED = False  # --aligned
N_SCRATCH = 3  # f buffers the warm-up launches cycle through (their results are not used)


PREWARM_MS = 60.0  # --prewarm-ms


def timed_run(M, torch, dist_mod, world, d, fs, steps, warmup):
    """fs: min(warmup, N_SCRATCH) scratch buffers for the warm-up launches, followed by one
    pristine f buffer per timed step (the routine works in place)."""
    def step(f):
        M.advect_scalar2D(f, d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"])

    if PREWARM_MS > 0:   # GPU wake-up on a private scratch copy (not one of the timed buffers)
        scratch = torch.empty_like(fs[-1]) if ALIGNED else M.empty_staggered(fs[-1].shape, "f", fs[-1].dtype, fs[-1].device)
        scratch.copy_(fs[-1])
        t_end = time.perf_counter() + PREWARM_MS * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(8):
                step(scratch)
            torch.cuda.synchronize()
        del scratch
    nscr = min(warmup, N_SCRATCH)
    for i in range(warmup):
        step(fs[i % nscr])
    fs = fs[nscr:]
    warmup = 0
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        step(fs[warmup + i])
        ev[i][1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist_mod.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = [a.elapsed_time(b) for a, b in ev]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist_mod.get_backend() == "nccl" else "cpu")
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
        dt = float(t.item())
    return dt, kms


def main():
    global ALIGNED, PREWARM_MS
    args = parse()
    ALIGNED = args.aligned
    PREWARM_MS = args.prewarm_ms
    import torch
    import torch.distributed as dist
    import codesign_kernels_amd as M

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # MPDATA_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend -- exercises the N > 1
    # control flow on a one-GPU box (RCCL refuses two ranks on one device); numbers are meaningless
    rehearsal = os.environ.get("MPDATA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    M.set_variant(M.VARIANT_FAST if args.variant == "fast" else M.VARIANT_EXACT)
    M.set_tile(args.tile)
    nx, nz = args.nx, args.nz
    n_loc = args.ncrms_per_gpu
    n_glob = n_loc * world
    sl0 = rank * n_loc
    steps, warmup = args.steps, args.warmup

    # ---- headline: 1 tracer (or --tracers) ---------------------------------
    ntr = args.tracers
    f32 = args.dtype == "f32"
    tdt = torch.float32 if f32 else torch.float64
    d, fs = make_problem(M, torch, dev, n_loc, n_glob, sl0, nx, nz, ntr, steps + min(warmup, N_SCRATCH), args.dist, tdt)
    dt, kms = timed_run(M, torch, dist, world, d, fs, steps, warmup)
    cells_per_step = n_glob * nx * (nz - 1) * ntr
    value = cells_per_step * steps / dt
    alg_bytes = M.algorithmic_bytes(n_loc, nx, nz, ntr, f32=f32)  # per launch (one GPU)
    k_avg = sum(kms) / len(kms)
    achieved = alg_bytes / (k_avg * 1e-3) / 1e9
    del fs
    torch.cuda.empty_cache()

    result = None
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{args.variant}_ncrms{n_loc}_nx{nx}_nz{nz}_t{ntr}" + ("_f32" if f32 else "")
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "advected cell-updates/sec, MPDATA advect_scalar2D (ncrms=65536 per GPU, nx=32, nz=28)",
            "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[2]: ncrms={n_loc}/GPU (global {n_glob}) nx={nx} "
                                   f"nz={nz} {'fp32 (NOT the headline precision)' if f32 else 'fp64'} "
                                   f"tracers={ntr}, device-resident, in-place f",
                       "ncrms_per_gpu": n_loc, "ncrms_global": n_glob, "nx": nx, "nz": nz,
                       "ntracers": ntr, "variant": args.variant, "input_law": args.dist,
                       "placement": "aligned" if args.aligned else "f,u,w staggered mod 1 KiB",
                       "prewarm_ms": args.prewarm_ms,
                       "parallelism": f"ncrms-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms_avg": k_avg, "kernel_ms_min": min(kms),
                         "kernel_ms_median": statistics.median(kms),
                         "cell_updates_per_sec_kernel": n_loc * nx * (nz - 1) * ntr / (k_avg * 1e-3)},
        }

    if rank == 0:
        try:
            result["roofline"]["measured_copy_GBs"] = copy_ceiling(torch, dev)
        except Exception:
            result["roofline"]["measured_copy_GBs"] = None

    # ---- side measurement: tracer-batched variant (configs[3]/[4]) -----------
    if not args.no_batched and ntr == 1:
        bt = args.batched_tracers
        bsteps, bwarm = min(steps, 5), min(warmup, 2)
        d2, fs2 = make_problem(M, torch, dev, n_loc, n_glob, sl0, nx, nz, bt, bsteps + min(bwarm, N_SCRATCH), args.dist, tdt)
        dt2, kms2 = timed_run(M, torch, dist, world, d2, fs2, bsteps, bwarm)
        if rank == 0:
            ab = M.algorithmic_bytes(n_loc, nx, nz, bt, f32=f32)
            ka = sum(kms2) / len(kms2)
            result["tracer_batched"] = {
                "workload": f"BASELINE.json configs[3]: ncrms={n_loc}/GPU, {bt} tracers sharing u,w,rho,rhow,adz",
                "value": n_glob * nx * (nz - 1) * bt * bsteps / dt2, "unit": "cell-updates/s",
                "steps": bsteps, "ms_per_step": dt2 / bsteps * 1e3,
                "roofline": {"bound": "hbm", "achieved": ab / (ka * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": ab / (ka * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": ab, "kernel_ms_avg": ka}}
        del fs2, d2
        torch.cuda.empty_cache()

    # ---- side measurement: the same workload in fp32 (reference precision switch) ------
    if not args.no_fp32 and not f32 and ntr == 1:
        d3, fs3 = make_problem(M, torch, dev, n_loc, n_glob, sl0, nx, nz, 1, steps + min(warmup, N_SCRATCH), args.dist, torch.float32)
        dt3, kms3 = timed_run(M, torch, dist, world, d3, fs3, steps, warmup)
        if rank == 0:
            ab = M.algorithmic_bytes(n_loc, nx, nz, 1, f32=True)
            ka = sum(kms3) / len(kms3)
            result["fp32"] = {
                "workload": f"ncrms={n_loc}/GPU nx={nx} nz={nz} fp32, 1 tracer (mpdata_advect_scalar2d_f32_device)",
                "value": n_glob * nx * (nz - 1) * steps / dt3, "unit": "cell-updates/s", "steps": steps,
                "ms_per_step": dt3 / steps * 1e3,
                "roofline": {"bound": "hbm", "achieved": ab / (ka * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": ab / (ka * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": ab, "kernel_ms_avg": ka}}
        del fs3, d3
        torch.cuda.empty_cache()

    # ---- side measurement: the second kernel, rank 0 only -------------------
    if not args.no_bwk and rank == 0 and not f32 and ntr == 1:
        try:
            result["biharmonic_wk"] = bench_bwk(torch, dev, min(steps, 50), min(warmup, 50),
                                                world == 1 and not args.no_cpu_baseline)
        except Exception as exc:   # a side measurement must not take the headline down
            result["biharmonic_wk"] = {"error": repr(exc)}
        try:
            result["high_order_flux"] = bench_nlk(torch, dev, min(steps, 100), min(warmup, 20),
                                                  world == 1 and not args.no_cpu_baseline)
        except Exception as exc:
            result["high_order_flux"] = {"error": repr(exc)}
        torch.cuda.empty_cache()

    # ---- optional: scatter/gather over RCCL (outside any timed region) ------
    if args.scatter and world > 1:
        ns = 4096 * world
        names = ("adz", "f", "u", "w", "rho", "rhow", "flux")
        sh = M.shapes(ns, nx, nz)
        if rank == 0:
            full = {k: torch.empty(sh[k], dtype=torch.float64, device=dev) for k in names}
            for k in names:
                M.fill_synthetic(full[k], k, 100, 1)
            arg = full
        else:
            arg = {k: sh[k][:-1] for k in names}
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        mine = M.scatter_inputs(arg, ns, src=0, device=dev)
        torch.cuda.synchronize(); dist.barrier()
        t_sc = time.perf_counter() - t0
        M.advect_scalar2D(mine["f"], mine["u"], mine["w"], mine["rho"], mine["rhow"], mine["flux"], mine["adz"])
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        M.gather_outputs({"f": mine["f"], "flux": mine["flux"]}, full if rank == 0 else None, ns, dst=0)
        torch.cuda.synchronize(); dist.barrier()
        t_ga = time.perf_counter() - t0
        if rank == 0:
            result["scatter_gather"] = {"ncrms": ns, "scatter_s": t_sc, "gather_s": t_ga}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(nx, nz)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
