#!/usr/bin/env python3
"""bench.py -- MPDATA tracer-advection throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU.  Either torch.distributed.run starts the ranks -- WORLD_SIZE is then
     set -- or the plain command does it itself: before torch is imported and before anything has
     touched a GPU it starts `python -m torch.distributed.run ... bench.py <same arguments>` as a
     CHILD process, passes its output through and returns its exit code, see self_launch())

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
    ap.add_argument("--prewarm-ms", type=float, default=500.0,
                    help="CAP of the untimed GPU wake-up before the warm-up steps: scratch launches of the same kernel "
                         "on cold field sets until their time has settled (plateau rule, wake_up()) or this long, so "
                         "that the clock / power-state ramp after idle (DESIGN.md 7) is over even when --warmup is "
                         "small; 0 disables")
    ap.add_argument("--prewarm-min-ms", type=float, default=40.0,
                    help="the wake-up lasts at least this long (the first launches after idle run fast, the power "
                         "management then takes the clock down: three equal groups inside that phase are no plateau)")
    ap.add_argument("--block-timeout", type=float, default=float(os.environ.get("MPDATA_BENCH_BLOCK_TIMEOUT", "420")),
                    help="seconds a side block (set-up, timing and the ranks' agreement after it) may take before the "
                         "watchdog ends the run with the line printed (class Lifeline); 0 disables")
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
    ap.add_argument("--no-exact", action="store_true", help="skip the exact_variant side block (EXACT: plan run, 25 tracers, device call)")
    ap.add_argument("--aligned", action="store_true",
                    help="allocate f, u, w with equally aligned bases (plain torch.empty) instead of the "
                         "staggered placement (DESIGN.md 4.3: 8 %% slower at ncrms=65536)")
    ap.add_argument("--tile", type=int, default=-1)
    ap.add_argument("--dist", type=int, default=1, help="1 conditioned, 2 reference-raw, 3 raw-signed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the 25-tracer side measurement")
    ap.add_argument("--batched-tracers", type=int, default=25)
    ap.add_argument("--batched-steps", type=int, default=20, help="timed steps of the 25-tracer block (at most --steps)")
    ap.add_argument("--no-scatter", action="store_true", help="N > 1: skip the scatter/gather measurement")
    ap.add_argument("--no-host-call", action="store_true",
                    help="skip the end-to-end block (the drop-in call on host arrays, PCIe-inclusive)")
    ap.add_argument("--no-reflayout", action="store_true",
                    help="skip the side measurement of the reference-layout device call (x-march kernel)")
    ap.add_argument("--serpentine", action="store_true",
                    help="serpentine tile order (DESIGN.md 4.1): off by default, only pays with --shared-uw")
    ap.add_argument("--shared-uw", action="store_true",
                    help="headline on ONE plan whose field sets share u, w (round-2 protocol); default: a plan per "
                         "field set with its own u, w")
    ap.add_argument("--no-fresh-uw", action="store_true", help="skip the step_with_fresh_uw side block")
    ap.add_argument("--no-x2", action="store_true", help="skip the twice_the_instances side block")
    ap.add_argument("--no-shared-block", action="store_true", help="skip the shared-u,w side block")
    ap.add_argument("--headline-only", action="store_true",
                    help="the headline block and the cpu_baseline only: no side block runs")
    ap.add_argument("--layout", choices=["wavemajor", "reference"], default="wavemajor",
                    help="device layout of the plans the headline runs on (include/mpdata_hip.h section 3)")
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
    nest, FAST variant, device-resident.  Two sizes: the reference's namelist
    (nested_loops/nested.nml: 25600 edges, 2800 cells, 100 levels, 10 cells per edge: a 37-us launch
    on a 69-MB working set, i.e. launch / L2-gather territory -- reported WITHOUT an HBM fraction)
    and a mesh 32 x larger (819200 edges, 89600 cells: 2.2 GB of compulsory traffic per call, well
    past the 256-MB Infinity Cache), where the HBM roofline is the right yardstick."""
    import numpy as np
    import codesign_kernels_amd.nlk as K
    K.set_variant(K.VARIANT_FAST)
    coef = float(np.float32(2.14))

    def one(nE, nC, nV, nA, steps, warmup, window=0):
        """window = 0: the reference's law, every edge draws its cells from the WHOLE mesh
        (nested.F90:84-90 "Create a random connectivity"); window = W: the cells of edge e lie within
        W cells of e * nCells / nEdges -- what an ordered unstructured mesh looks like (an edge's
        advection cells are its two cells and their neighbours)."""
        g = torch.Generator(device=dev).manual_seed(3)
        rnd = lambda *shape: torch.rand(shape, dtype=torch.float64, device=dev, generator=g)
        if window:
            c0 = (torch.arange(nE, device=dev, dtype=torch.int64) * nC // nE).view(nE, 1)
            off = torch.randint(-window, window + 1, (nE, nA), device=dev, generator=g)
            cells = (torch.clamp(c0 + off, 0, nC - 1) + 1).to(torch.int32)
        else:
            cells = torch.randint(1, nC + 1, (nE, nA), dtype=torch.int32, device=dev, generator=g)
        d = {"nAdvCellsForEdge": torch.full((nE,), nA, dtype=torch.int32, device=dev),
             "advCellsForEdge": cells,
             "minLevelCell": torch.ones((nC,), dtype=torch.int32, device=dev),
             "maxLevelCell": torch.clamp((rnd(nC) * nV * 2).round().to(torch.int32), 3, nV),
             "tracerCur": 15.0 * rnd(nC, nV), "normalThicknessFlux": 15.0 * (0.5 - rnd(nE, nV)),
             "advMaskHighOrder": torch.ones((nE, nV), dtype=torch.float64, device=dev),
             "advCoefs": 20.0 * rnd(nE, nA), "advCoefs3rd": 21.0 * rnd(nE, nA)}
        out = torch.zeros((nE, nV), dtype=torch.float64, device=dev)
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
        return dt, kms, K.algorithmic_bytes(nE, nC, nV, nV, nA)

    nE, nC, nV, nA = 25600, 2800, 100, 10
    dt, kms, ab = one(nE, nC, nV, nA, steps, warmup)
    res = {"workload": f"nested_loops/nested.F90 high-order flux loop nest: nEdges={nE} nCells={nC} "
                       f"nVertLevels={nV} nAdv={nA} fp64, device-resident (the reference's namelist size)",
           "value": nE * nV * steps / dt, "unit": "edge-level fluxes/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
           "kernel_ms_avg": kms, "algorithmic_bytes_per_launch": ab,
           "note": "69 MB per call, inside the Infinity Cache, 37 us per launch: not an HBM-bound measurement, no "
                   "roofline fraction is claimed for this size"}
    big = 32
    s2 = max(5, steps // 5)
    # (a) an ORDERED mesh: the cells of an edge within +-128 cells of the edge's position -- the gathered
    #     columns of neighbouring edges overlap and stay in L2; compulsory bytes are the yardstick
    dt2, kms2, ab2 = one(nE * big, nC * big, nV, nA, s2, 3, window=128)
    res["large_mesh"] = {
        "workload": f"the same nest on a mesh {big} x larger with LOCAL connectivity (cells of an edge within 128 cells "
                    f"of its position, as in an ordered mesh): nEdges={nE * big} nCells={nC * big} nVertLevels={nV} nAdv={nA}",
        "value": nE * big * nV * s2 / dt2, "unit": "edge-level fluxes/s", "steps": s2, "ms_per_step": dt2 / s2 * 1e3,
        "roofline": {"bound": "hbm", "achieved": ab2 / (kms2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ab2 / (kms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab2,
                     "kernel_ms_avg": kms2, "note": "compulsory bytes (every array once); gathers served by L2"}}
    # (b) the reference's own law at this size: every edge draws its 10 cells from the whole 72-MB tracer
    #     table -- no blocking of edges can make those gathers local; each is an 800-byte read that misses
    #     L2 (4 MB per XCD) and is served by the Infinity Cache / HBM
    dt3, kms3, ab3 = one(nE * big, nC * big, nV, nA, s2, 3)
    gather = nE * big * nA * nV * 8
    res["large_mesh_random_connectivity"] = {
        "workload": f"mesh {big} x larger with the reference's RANDOM connectivity (nested.F90:84-90): the 10 gathered "
                    f"columns of an edge are random rows of a {nC * big * nV * 8 / 1e6:.0f}-MB table",
        "value": nE * big * nV * s2 / dt3, "unit": "edge-level fluxes/s", "steps": s2, "ms_per_step": dt3 / s2 * 1e3,
        "roofline": {"bound": "hbm", "achieved": ab3 / (kms3 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ab3 / (kms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ab3,
                     "kernel_ms_avg": kms3,
                     "gather_bytes_per_launch": gather,
                     "rate_incl_gathers_GBs": (ab3 + gather) / (kms3 * 1e-3) / 1e9,
                     "note": "frac counts compulsory bytes only; the launch additionally gathers "
                             f"{gather / 1e9:.1f} GB of tracerCur rows that cannot hit in L2 under this law: "
                             "rate_incl_gathers_GBs is what the memory system delivers"}}
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


def make_shared(M, torch, dev, ncrms_loc, ncrms_glob, sl0, nx, nz, dist, dtype):
    """u, w, rho, rhow, adz, flux of one rank in the reference layout, generated on the device."""
    sh = M.shapes(ncrms_loc, nx, nz, 1)
    alloc = (lambda shape, k: torch.empty(shape, dtype=dtype, device=dev)) if ALIGNED else \
            (lambda shape, k: M.empty_staggered(shape, k, dtype, dev))
    d = {k: alloc(sh[k], k) for k in ("u", "w", "rho", "rhow", "adz", "flux")}
    for k in d:
        M.fill_synthetic(d[k], k, 100, dist, ncrms_global=ncrms_glob, sl0=sl0)
    return d, alloc, sh


ALIGNED = False  # --aligned (reference-layout side measurement only)
N_SCRATCH = 3    # field buffers the warm-up launches cycle through (their results are not used)
PREWARM_MS = 500.0     # --prewarm-ms (cap of the plateau rule)
PREWARM_MIN_MS = 40.0  # --prewarm-min-ms
# VALU instructions one wave executes per call at nx=32, nz=28 (rocprofv3 SQ_INSTS_VALU / SQ_WAVES,
# profiles/r02_pmc_summary.json) and the measured fp64 VALU issue peak of the chip
# (tools/valu_rate.hip: 33e12 lane-ops/s = 515.6e9 wave-instructions/s)
def valu_instructions_per_launch(variant, n_loc, nx, nz, ntr):
    """VALU wave-instructions of one plan run of this shape, from the SQ counters of the recorded
    profile (profiles/*_pmc_summary.json: t25_wavemajor.valu_instructions_per_launch); None if the
    shape was not profiled."""
    if variant != "fast" or (n_loc, nx, nz, ntr) != (65536, 32, 28, 25):
        return None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            with open(path) as fh:
                v = json.load(fh).get("t25_wavemajor", {}).get("valu_instructions_per_launch")
            if v:
                return float(v), os.path.relpath(path, ROOT)
        except Exception:
            pass
    return None


VALU_PEAK_WAVE_INSTR_PER_S = 33.0e12 / 64.0
# The side blocks (reference-layout call, fp32, second / third kernel) are steady-state measurements of
# their own: a fixed warm-up, whatever --warmup the headline was given (the driver's W = 5 left them in the
# first-launches transient: 4-6 % slow)
SIDE_WARMUP = 30


def free_bytes(torch):
    return torch.cuda.mem_get_info()[0]


class InjectedFailure(RuntimeError):
    """MPDATA_BENCH_FAIL=<block>:<rank>[:mid][,...] (tests/test_bench_rehearsal.py): the named block raises
    on that rank, at its start or in the middle of its timed loop"""


FAIL_SPEC = os.environ.get("MPDATA_BENCH_FAIL", "")
CURRENT_BLOCK = [None]   # name of the side block that is running (failure injection)


def inject_failure(name, phase):
    for spec in FAIL_SPEC.split(","):
        parts = spec.split(":")
        if len(parts) < 2 or parts[0] != name or int(parts[1]) != int(os.environ.get("RANK", "0")):
            continue
        how = parts[2] if len(parts) > 2 else "start"
        if how == "hang" and phase == "start":     # a rank that never comes back (a hung collective, a stuck kernel)
            while True:
                time.sleep(1.0)
        if how == "kill" and phase == "start":     # a rank that dies without a word (SIGKILL: out of memory, a GPU fault)
            os.kill(os.getpid(), 9)
        if how == phase:
            raise InjectedFailure(f"injected failure in block {name!r} ({phase}) on rank {parts[1]}")


# The driver keeps the last 10 KB of the line for its reader: the side kernels and the secondary views go first, the
# blocks of the hot path (and the EXACT variant a default caller gets) last -- JSON object order carries no meaning
TAIL_LAST = ("cpu_baseline", "twice_the_instances", "two_launches_in_flight", "reference_layout_device_call",
             "step_with_fresh_uw", "exact_variant", "tracer_batched", "scatter_gather")
HEAD_AFTER_HEADLINE = ("layout_conversion", "biharmonic_wk", "high_order_flux", "end_to_end_host_call", "fp32",
                       "consecutive_tracers_shared_uw", "levels_above_64")


def ordered_for_the_tail(d):
    core = [k for k in d if k not in TAIL_LAST and k not in HEAD_AFTER_HEADLINE]
    keys = core + [k for k in HEAD_AFTER_HEADLINE if k in d] + [k for k in TAIL_LAST if k in d]
    out = {k: d[k] for k in keys}
    # ... and the headline once more as the LAST key (a reader of the tail sees it next to the hot-path blocks)
    r = d.get("roofline") or {}
    out["headline_repeated"] = {"metric": d.get("metric"), "value": d.get("value"), "unit": d.get("unit"), "n_gpus": d.get("n_gpus"),
                                "ms_per_step": d.get("ms_per_step"), "roofline_frac": r.get("frac"),
                                "kernel_ms_avg": r.get("kernel_ms_avg")}
    return out


class Lifeline:
    """What no try / except can catch.  (a) A block that never ends -- a collective that hangs because a rank is
    gone, a kernel that does not finish: a thread on every rank watches the deadline of the running block; when it
    passes, rank 0 records an error entry under the block's name, prints THE line with everything measured so far
    and every rank leaves with exit code 0.  (b) SIGTERM -- what torch.distributed.run sends to the surviving ranks
    when one rank died, and what a `timeout` sends: the main thread may sit in a collective then and never reach a
    Python-level handler, so the signal is taken off CPython's wake-up descriptor (the C-level handler writes the
    signal number there whatever the main thread is doing) by a thread of its own; rank 0 prints the line, then the
    process exits with 143.  The line goes out once, whoever prints it."""

    def __init__(self):
        import signal
        import threading
        self.lock = threading.Lock()
        self.printed = False
        self.result = None          # rank 0: the dict of the line, once the headline exists
        self.rank = int(os.environ.get("RANK", "0"))
        self.block, self.deadline, self.seconds = None, None, 0.0
        self.rfd, wfd = os.pipe()
        os.set_blocking(wfd, False)
        signal.signal(signal.SIGTERM, lambda signum, frame: None)   # (a Python-level handler arms the C-level one)
        signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)
        self.sigterm = int(signal.SIGTERM)
        threading.Thread(target=self._sigterm, daemon=True).start()
        threading.Thread(target=self._watch, daemon=True).start()

    def print_line(self, extra=None):
        with self.lock:
            if self.printed or self.rank != 0 or self.result is None:
                return
            self.printed = True
            for _ in range(3):      # (the main thread may be writing into the dict)
                try:
                    line = json.dumps(ordered_for_the_tail(dict(self.result, **(extra or {}))))
                    break
                except RuntimeError:
                    time.sleep(0.05)
            else:
                line = json.dumps(extra or {})
            sys.stdout.write(line + "\n")
            sys.stdout.flush()

    def arm(self, block, seconds):
        self.block, self.seconds = block, seconds
        self.deadline = time.monotonic() + seconds if seconds > 0 else None

    def disarm(self):
        self.deadline, self.block = None, None

    def _sigterm(self):
        while True:
            b = os.read(self.rfd, 1)
            if b and b[0] == self.sigterm:
                break
        blk = self.block
        self.print_line({"terminated": "SIGTERM" + (f" during block {blk!r}" if blk else "") +
                         " (another rank died?): the line holds what was measured until then"})
        os._exit(143)

    def _watch(self):
        while True:
            time.sleep(0.5)
            dl, blk = self.deadline, self.block
            if dl is None or time.monotonic() < dl:
                continue
            msg = f"watchdog: block {blk!r} had not ended after {self.seconds:.0f} s (a hang or a lost rank); the run was ended here"
            if self.rank == 0:
                print("BENCH_WATCHDOG " + msg, file=sys.stderr, flush=True)
                self.print_line({blk: {"error": msg}} if blk else {"watchdog": msg})
            else:
                time.sleep(2.0)     # (rank 0 first)
            os._exit(0)


def wake_up(torch, prewarm_launch, cap_ms, min_ms):
    """GPU wake-up before the warm-up steps, PLATEAU rule: scratch launches of the block's own kernel on cold field
    sets, in groups of 8 with one HIP-event pair per group, until three consecutive groups agree within 1 % (and at
    least `min_ms` have passed: the first launches after idle run FAST, then the power management takes the clock
    down) or `cap_ms` have passed.  A fixed 60 ms (round 4) was over before the clock had settled on a
    fresh box: the 20 timed steps of the driver's command (8 ms) then sat on the ramp.  The reference's own protocol
    is the same idea -- the first call pays the warm-up, the second is quoted (results/advect.pgiacc.17.7:2-13)."""
    info = {"rule": "groups of 8 cold launches until 3 consecutive group means agree within 1 %",
            "cap_ms": cap_ms, "min_ms": min_ms, "ms_used": 0.0, "groups": 0, "plateau_reached": None,
            "group_ms_per_launch": []}
    if cap_ms <= 0 or prewarm_launch is None:
        return info
    t0 = time.perf_counter()
    n, groups = 0, []
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            prewarm_launch(n)
            n += 1
        e1.record()
        torch.cuda.synchronize()
        groups.append(e0.elapsed_time(e1) / 8)
        used = (time.perf_counter() - t0) * 1e3
        last = groups[-3:]
        if len(last) == 3 and used >= min_ms and max(last) - min(last) <= 0.01 * min(last):
            info["plateau_reached"] = True
            break
        if used >= cap_ms:
            info["plateau_reached"] = False
            break
    info.update(ms_used=used, groups=len(groups), group_ms_per_launch=[round(g, 5) for g in groups[-12:]])
    return info


def timed_loop(torch, dist_mod, world, launch, steps, warmup, prewarm_launch=None, collective=True):
    """W untimed warm-up steps, then EXACTLY `steps` timed steps bracketed by barrier +
    synchronize, with ONE pair of HIP events on the launch stream around the K launches (kernel
    time per step = elapsed / K: the kernels run back to back, nothing else is on the stream; an
    event pair per step would put two marker packets between any two kernels and is kept out of
    the timed region).  A second, untimed pass with an event pair per step gives the spread.
    Returns (max-over-ranks wall seconds, [mean kernel ms per step] + per-step samples).

    collective = False (every side block): NO barrier and no all-reduce in here -- the rank times
    its own K steps, and main() takes the maximum over the ranks in the one all-reduce that every
    rank reaches whether its block raised or not (a rank that raises between two barriers would
    leave the others waiting until the driver's timeout)."""
    wake = wake_up(torch, prewarm_launch, PREWARM_MS, PREWARM_MIN_MS)
    if world > 1 and collective:
        # the ranks reach their plateaus at different times; the ones that were first would sit idle in the barrier
        # in front of the timed region and start it on the clock ramp again.  So: meet here, then a SHORT second
        # wake-up (same rule, at most 100 ms) that every rank starts at the same moment
        torch.cuda.synchronize()
        dist_mod.barrier()
        wake["after_barrier"] = wake_up(torch, prewarm_launch, min(PREWARM_MS, 100.0), 0.0)
    for i in range(warmup):
        launch(-1 - i)
    if CURRENT_BLOCK[0] is not None:
        inject_failure(CURRENT_BLOCK[0], "mid")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if world > 1 and collective:
        dist_mod.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        launch(i)
    e1.record()
    torch.cuda.synchronize()
    if world > 1 and collective:
        dist_mod.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_avg = e0.elapsed_time(e1) / steps
    # spread: the same launches once more, one event pair each (not timed, not part of `value`)
    ns = min(steps, 20)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ns)]
    for i in range(ns):
        ev[i][0].record()
        launch(i)
        ev[i][1].record()
    torch.cuda.synchronize()
    samples = [a.elapsed_time(b) for a, b in ev]
    if world > 1 and collective:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist_mod.get_backend() == "nccl" else "cpu")
        dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
        dt = float(t.item())
    return dt, KernelTimes(k_avg, samples, wake)


class KernelTimes(list):
    """per-step event samples (list) + the mean over the timed region (`avg`) + what the wake-up did"""
    def __init__(self, avg, samples, wake=None):
        super().__init__(samples)
        self.avg = avg
        self.wake = wake


MAX_COLD_SETS = 48   # distinct field sets the timed steps cycle through (each 1.6 GB at configs[2]: far past the 256-MB Infinity Cache)


def _ev_ms(torch, fn, reps=1):
    """milliseconds of fn() on the current stream (HIP events), mean of `reps` back-to-back calls"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


class PlanSets:
    """The field sets one block steps through behind the plan API (device state in the library's own
    layout, include/mpdata_hip.h section 3): one step = one run of `ntr` tracers on a field set of its own.

    shared_uw = False (the headline): every field set is a PLAN OF ITS OWN with its own f AND its own
    u, w, rho, rhow, adz (same input law, per-set seed), so that no timed launch re-reads a byte the
    previous launch touched -- the reference's timed region is one call on its own data (:109-110,
    :237-239).  shared_uw = True: one plan, the field sets are tracers of it and share u, w (what
    consecutive tracers of one CRM step look like; 1.05 of the 2.14 GB of a launch are then the
    previous launch's bytes).  As many sets as the steps need are created if they fit in `mem_frac`
    of the free memory (at most MAX_COLD_SETS distinct ones); otherwise the timed steps cycle.

    Creating the sets (the allocations: the likeliest thing to fail) is separate from timing them, so
    that the ranks of a multi-GPU run can agree that every one of them got this far before the first
    barrier of the timed loop."""

    def __init__(self, M, torch, dev, shared, shape_f, n_loc, n_glob, sl0, nx, nz, ntr, steps, warmup,
                 dist_law, np_dtype, tdt, mem_frac=0.55, shared_uw=False):
        self.plans = []
        try:
            self._setup(M, torch, dev, shared, shape_f, n_loc, n_glob, sl0, nx, nz, ntr, steps, warmup, dist_law,
                        np_dtype, tdt, mem_frac, shared_uw)
        except BaseException:
            self.close(torch)
            raise

    def _setup(self, M, torch, dev, shared, shape_f, n_loc, n_glob, sl0, nx, nz, ntr, steps, warmup, dist_law,
               np_dtype, tdt, mem_frac, shared_uw):
        eb = 8 if tdt == torch.float64 else 4
        f_bytes = n_loc * (nx + 6) * (nz - 1) * eb * ntr
        uw_bytes = n_loc * ((nx + 5) * (nz - 1) + (nx + 4) * nz) * eb
        set_bytes = f_bytes + (0 if shared_uw else int(1.1 * uw_bytes))
        want = steps + min(warmup, N_SCRATCH)
        if not shared_uw:
            want = min(want, MAX_COLD_SETS)
        nset = int(max(2, min(want, (free_bytes(torch) * mem_frac) // set_bytes)))
        ftmp = torch.empty(shape_f, dtype=tdt, device=dev)
        conv = {}
        plans = self.plans
        if shared_uw:
            plan = M.Plan(n_loc, nx, nz, nset * ntr, dtype=np_dtype)
            plans.append(plan)
            plan.set_stream()
            plan.set_timing(False)   # (timed_loop brackets the K launches with ONE event pair)
            plan.import_device(None, shared["u"], shared["w"], shared["rho"], shared["rhow"], shared["adz"], None)
            for t in range(nset * ntr):     # per-tracer / per-set seeds: distinct data, same law
                M.fill_synthetic(ftmp, "f", 100 + t, dist_law, ncrms_global=n_glob, sl0=sl0)
                plan.import_device(ftmp, flux=shared["flux"], first_tracer=t)
            run = lambda s: plan.run(s * ntr, ntr)
        else:
            tmp = {k: torch.empty_like(shared[k]) for k in ("u", "w", "rho", "rhow", "adz")}
            for sset in range(nset):
                pl = M.Plan(n_loc, nx, nz, ntr, dtype=np_dtype)
                plans.append(pl)
                pl.set_stream()
                pl.set_timing(False)   # (timed_loop brackets the K launches with ONE event pair)
                for k in tmp:
                    M.fill_synthetic(tmp[k], k, 100 + 7919 * (sset + 1), dist_law, ncrms_global=n_glob, sl0=sl0)
                if sset == 0:   # layout-entry cost, measured on its own (no fill inside): u + w, then one tracer of f
                    pl.import_device(None, tmp["u"], tmp["w"], tmp["rho"], tmp["rhow"], tmp["adz"], None)
                    conv["import_ms_u_and_w"] = _ev_ms(torch, lambda: pl.import_device(None, tmp["u"], tmp["w"]), 3)
                pl.import_device(None, tmp["u"], tmp["w"], tmp["rho"], tmp["rhow"], tmp["adz"], None)
                for t in range(ntr):
                    M.fill_synthetic(ftmp, "f", 100 + sset * ntr + t, dist_law, ncrms_global=n_glob, sl0=sl0)
                    pl.import_device(ftmp, flux=shared["flux"], first_tracer=t)
                if sset == 0:
                    conv["import_ms_f_per_tracer"] = _ev_ms(torch, lambda: pl.import_device(ftmp, first_tracer=ntr - 1), 3)
                    conv["export_ms_f_per_tracer"] = _ev_ms(torch, lambda: pl.export_device(ftmp, first_tracer=ntr - 1), 3)
                    conv["bytes_per_array"] = int(ftmp.numel() * eb)
                    pl.import_device(ftmp, first_tracer=ntr - 1)   # (the export wrote into ftmp: same values back)
            del tmp
            run = lambda s: plans[s].run()
        torch.cuda.synchronize()
        del ftmp
        nscr = min(max(warmup, 1), N_SCRATCH, nset - 1)
        ntimed = nset - nscr
        self.run, self.nset, self.nscr, self.ntimed = run, nset, nscr, ntimed
        self.info = {"layout": "wave-major (plan-private)" if plans[0].layout == M.LAYOUT_WAVEMAJOR else "reference",
                     "field_sets": ntimed, "steps_per_field_set": -(-steps // ntimed),
                     "uw_shared_across_steps": bool(shared_uw), "conversion": conv}

    def launch(self, i):
        """i >= 0: timed step i on field set i mod ntimed; i < 0: warm-up step on the scratch sets"""
        self.run((self.nset - 1 - ((-1 - i) % self.nscr)) if i < 0 else (i % self.ntimed))

    def prewarm(self, n):
        """GPU wake-up launch number n: CYCLES through the scratch sets, so that consecutive wake-up
        launches work on different field sets as the timed ones do (1.6 GB each: nothing of a launch is
        left in the 256-MB Infinity Cache when its set comes round again) -- every dispatch of a
        profiled run is then a cold one and the plain average of its kernel trace is the cold figure"""
        self.launch(-1 - (n % self.nscr))

    def close(self, torch):
        for pl in self.plans:
            try:
                pl.close()
            except Exception:
                pass
        self.plans = []
        try:
            torch.cuda.empty_cache()
        except Exception:
            pass


def bench_plan(M, torch, dist_mod, world, dev, shared, shape_f, n_loc, n_glob, sl0, nx, nz, ntr, steps, warmup,
               dist_law, np_dtype, tdt, mem_frac=0.55, shared_uw=False, collective=False):
    """field sets + timed loop of a SIDE block (no collective inside: see timed_loop)"""
    sets = PlanSets(M, torch, dev, shared, shape_f, n_loc, n_glob, sl0, nx, nz, ntr, steps, warmup, dist_law, np_dtype,
                    tdt, mem_frac, shared_uw)
    try:
        dt, kms = timed_loop(torch, dist_mod, world, sets.launch, steps, warmup, prewarm_launch=sets.prewarm,
                             collective=collective)
        return dt, kms, sets.info
    finally:
        sets.close(torch)


def bench_fresh_uw(M, torch, dist_mod, world, dev, shared, shape_f, n_loc, n_glob, sl0, nx, nz, steps, warmup,
                   dist_law, np_dtype, tdt, nsets=16):
    """One step = mpdata_plan_run_uw(plan, u, w) with u, w in the REFERENCE layout on the device,
    distinct arrays for consecutive steps (a set = a plan with its own f + its own reference-layout
    u, w: 2.7 GB, the steps cycle through `nsets` of them).  Side block: no collective inside."""
    sets = []
    try:
        ftmp = torch.empty(shape_f, dtype=tdt, device=dev)
        for s_ in range(nsets):
            pl = M.Plan(n_loc, nx, nz, 1, dtype=np_dtype)
            sets.append((pl, None, None))
            pl.set_stream()
            pl.set_timing(False)
            # (u, w placed as every reference-layout array of this bench is: bases at different offsets modulo
            #  1 KiB -- INTEGRATION.md 6; two plain 2-MiB-aligned allocations put the row streams of u and w on
            #  the same HBM channels, 7 % slower)
            if ALIGNED:
                u = torch.empty_like(shared["u"]); w = torch.empty_like(shared["w"])
            else:
                u = M.empty_staggered(tuple(shared["u"].shape), "u", tdt, dev)
                w = M.empty_staggered(tuple(shared["w"].shape), "w", tdt, dev)
            M.fill_synthetic(u, "u", 500 + 31 * s_, dist_law, ncrms_global=n_glob, sl0=sl0)
            M.fill_synthetic(w, "w", 500 + 31 * s_, dist_law, ncrms_global=n_glob, sl0=sl0)
            M.fill_synthetic(ftmp, "f", 500 + s_, dist_law, ncrms_global=n_glob, sl0=sl0)
            pl.import_device(ftmp, u, w, shared["rho"], shared["rhow"], shared["adz"], shared["flux"])
            sets[-1] = (pl, u, w)
        del ftmp
        nscr = 2

        def launch(i):
            pl, u, w = sets[(nsets - 1 - ((-1 - i) % nscr)) if i < 0 else (i % (nsets - nscr))]
            pl.run_uw(u, w)

        # (the same wake-up as the headline's: the first ~30 ms of GPU activity after the set-up's idle phases run slow)
        dt, kms = timed_loop(torch, dist_mod, world, launch, steps, warmup, prewarm_launch=lambda n: launch(-1 - n),
                             collective=False)
        return dt, kms, nsets - nscr
    finally:
        for pl, _, _ in sets:
            try:
                pl.close()
            except Exception:
                pass
        del sets
        torch.cuda.empty_cache()


def kavg(kms):
    return kms.avg if isinstance(kms, KernelTimes) else sum(kms) / len(kms)


def roofline_block(alg_bytes, kms, extra=None, full=False):
    """full (the headline): with the per-launch list and what the wake-up did; the side blocks carry the digest only
    (the whole line stays near 12 KB: the driver keeps a 10-KB tail of it for the reader)"""
    k_avg = kavg(kms)
    ach = alg_bytes / (k_avg * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": None, "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_avg": k_avg,
         "kernel_ms_min": min(kms), "kernel_ms_median": statistics.median(kms),
         "avg_over_median": k_avg / statistics.median(kms)}
    wake = getattr(kms, "wake", None)
    if full:
        r["kernel_ms_note"] = "HIP events around the K timed launches / K; min / median / samples: event pair per launch, separate pass"
        r["kernel_ms_samples"] = [round(x, 4) for x in kms]
        if wake is not None:
            r["wake_up"] = wake
    elif wake is not None:
        r["wake_ms_used"] = round(wake["ms_used"], 1)
        r["wake_plateau_reached"] = wake["plateau_reached"]
    if extra:
        r.update(extra)
    return r


def traffic_lookup(key):
    """HBM bytes per launch from the PMC passes of tools/profile_round.sh (FETCH_SIZE x 2 +
    WRITE_SIZE as the guide prescribes), measured on the builder's box for exactly this kernel and
    size -- a recorded profile value, not a live measurement (`traffic_source` says so)."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        return json.load(open(tpath)).get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def self_launch(ngpus):
    """`python3 bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks
    ourselves, as the reference starts its whole run with one command (`./advect`, mmf-mpdata-tracer/README.md:20-21).
    This process has not imported torch and never touches a GPU: it starts `python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <a free port> bench.py <the same arguments>`
    as a CHILD process (never an exec: a process that has initialised the GPU must not be replaced, and the ranks
    are children of the launcher anyway), lets it write straight to our stdout / stderr, hands SIGTERM / SIGINT on
    to it and returns its exit code.  Nothing is retried: a failed attempt ends with the child's error output and
    its non-zero code."""
    import signal
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:   # a port the OS hands out
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (this pool's host driver: dmabuf IPC only)
    env["MPDATA_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print("BENCH_SELF_LAUNCH " + " ".join(cmd), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env)

    def forward(signum, frame):
        try:
            child.send_signal(signum)
        except OSError:
            pass
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, forward)
    while True:
        try:
            return child.wait()
        except KeyboardInterrupt:
            continue


def main():
    global ALIGNED, PREWARM_MS, PREWARM_MIN_MS
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    ALIGNED = args.aligned
    PREWARM_MS, PREWARM_MIN_MS = args.prewarm_ms, min(args.prewarm_min_ms, args.prewarm_ms)
    # (a rank started by someone else's launcher: the same default as self_launch gives its children, before HIP loads)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    life = Lifeline()
    import numpy as np
    import torch
    import torch.distributed as dist
    import codesign_kernels_amd as M

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:    # (WORLD_SIZE=1 set by hand; main() starts the ranks itself when it is unset)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1")
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

    # who is really there: ranks of the process group (= ncclCommCount of the RCCL communicator behind it)
    # and the HIP device every rank sits on
    ranks_seen, devices_seen = 1, [local]
    if world > 1:
        ranks_seen = dist.get_world_size()
        dv = torch.tensor([torch.cuda.current_device()], dtype=torch.int32, device="cpu" if rehearsal else dev)
        allv = [torch.zeros_like(dv) for _ in range(world)]
        dist.all_gather(allv, dv)
        devices_seen = [int(t.item()) for t in allv]
    M.set_variant(M.VARIANT_FAST if args.variant == "fast" else M.VARIANT_EXACT)
    M.set_tile(args.tile)
    serp = 1 if args.serpentine else 0
    M.set_serpentine(serp)
    if args.layout == "reference":
        M.set_plan_layout(M.LAYOUT_REFERENCE)
    nx, nz = args.nx, args.nz
    n_loc = args.ncrms_per_gpu
    n_glob = n_loc * world
    sl0 = rank * n_loc
    steps, warmup = args.steps, args.warmup
    mem_frac = 0.55 / (world if rehearsal else 1)

    ntr = args.tracers
    f32 = args.dtype == "f32"
    tdt = torch.float32 if f32 else torch.float64
    npdt = np.float32 if f32 else np.float64
    shared, alloc, sh = make_shared(M, torch, dev, n_loc, n_glob, sl0, nx, nz, args.dist, tdt)
    sh_f1 = M.shapes(n_loc, nx, nz, 1)["f"]
    cells_1 = n_glob * nx * (nz - 1)

    # ---- the ranks' agreement: ONE all-reduce that every rank reaches, whether its block raised or not
    def agree(failed, dt=0.0):
        """-> (any rank failed, max over ranks of dt)"""
        if world == 1:
            return bool(failed), float(dt)
        t = torch.tensor([1.0 if failed else 0.0, float(dt)], dtype=torch.float64,
                         device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t[0].item() > 0), float(t[1].item())

    def emit(obj, code=0):
        if rank == 0:
            print(json.dumps(obj), flush=True)
        if world > 1:
            try:
                dist.destroy_process_group()
            except Exception:
                pass
        if code:
            sys.exit(code)

    # ---- headline: configs[2] per GPU through the plan API -----------------------------------
    # phase 1, no collective: the field sets (the allocations); the ranks then agree that all of them got
    # this far BEFORE the first barrier of the timed loop
    sets, err = None, None
    try:
        sets = PlanSets(M, torch, dev, shared, sh_f1, n_loc, n_glob, sl0, nx, nz, ntr, steps, warmup, args.dist, npdt,
                        tdt, mem_frac, shared_uw=args.shared_uw)
    except Exception as exc:
        err = repr(exc)
    if agree(err is not None)[0]:
        emit({"metric": "advected cell-updates/sec, MPDATA advect_scalar2D", "value": None, "n_gpus": world,
              "error": "headline setup failed: " + (err or "on another rank")}, code=1)
        return
    # phase 2: W warm-up + K timed steps, barrier + synchronize on both sides, max over ranks
    try:
        dt, kms = timed_loop(torch, dist, world, sets.launch, steps, warmup, prewarm_launch=sets.prewarm, collective=True)
        info = sets.info
    finally:
        sets.close(torch)
    del sets
    value = cells_1 * ntr * steps / dt
    alg_bytes = M.algorithmic_bytes(n_loc, nx, nz, ntr, f32=f32)  # per launch (one GPU)

    result = {}
    if rank == 0:
        key = f"{args.variant}_ncrms{n_loc}_nx{nx}_nz{nz}_t{ntr}" + ("_f32" if f32 else "") + \
              ("_wm" if info["layout"].startswith("wave") else "")
        traffic = traffic_lookup(key)
        result = {
            "metric": f"advected cell-updates/sec, MPDATA advect_scalar2D (ncrms={n_loc} per GPU, nx={nx}, nz={nz})",
            "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[2]: ncrms={n_loc}/GPU (global {n_glob}) nx={nx} "
                                   f"nz={nz} {'fp32 (NOT the headline precision)' if f32 else 'fp64'} "
                                   f"tracers={ntr}, device-resident in a plan (include/mpdata_hip.h 3), in-place f",
                       "ncrms_per_gpu": n_loc, "ncrms_global": n_glob, "nx": nx, "nz": nz,
                       "ntracers": ntr, "variant": args.variant, "input_law": args.dist,
                       "device_layout": info["layout"], "field_sets": info["field_sets"],
                       "steps_per_field_set": info["steps_per_field_set"],
                       "uw_shared_across_steps": info["uw_shared_across_steps"],
                       "serpentine": bool(serp),
                       "cold": "every timed step runs on a plan of its own: own f, u, w, rho, rhow, adz (no byte of a "
                               "launch was touched by the previous one)" if not info["uw_shared_across_steps"] else
                               "NO: the field sets share u, w",
                       "prewarm_ms_cap": args.prewarm_ms, "prewarm_ms_used": kms.wake["ms_used"],
                       "prewarm_plateau_reached": kms.wake["plateau_reached"],
                       "prewarm": "wake-up launches cycle through the scratch field sets (cold, like the timed ones) "
                                  "until three consecutive groups of 8 agree within 1 % (roofline.wake_up)",
                       "launched_by": "bench.py itself (child torch.distributed.run)" if
                                      os.environ.get("MPDATA_BENCH_SELF_LAUNCHED") == "1" else
                                      ("torch.distributed.run" if world > 1 else "single process"),
                       "parallelism": f"ncrms-sharded x{world}, no data-path collective",
                       "ranks_seen": ranks_seen, "devices_seen": devices_seen},
            "roofline": roofline_block(alg_bytes, kms, full=True, extra={
                "traffic": traffic,
                "traffic_source": None if traffic is None else "profiles/hbm_traffic.json[%s]: PMC passes of "
                                  "tools/profile_round.sh on the builder's box, not measured in this run" % key,
                "cell_updates_per_sec_kernel": n_loc * nx * (nz - 1) * ntr / (kavg(kms) * 1e-3)}),
            "layout_conversion": dict(info["conversion"], note=
                                      "reference layout <-> plan layout on the device (HIP events, no fill inside), "
                                      "outside the timed region of the headline like the reference's `!$acc update "
                                      "device`, :107; `step_with_fresh_uw` charges it"),
        }
        # the headline exists: on stderr at once (a later block that takes the process down cannot lose it);
        # stdout gets the ONE full line at the end
        print("BENCH_HEADLINE " + json.dumps(result), file=sys.stderr, flush=True)
        life.result = result

    # ---- side blocks.  Each one is a closure that does its set-up AND its timing WITHOUT any collective and
    #      returns (seconds of its K steps on this rank, entry(max seconds over the ranks) -> dict).  An
    #      exception on any rank becomes {"error": ...} under the block's name; every rank then meets in the
    #      one all-reduce of side() and goes on to the next block.
    def side(name, body, rank0_only=False):
        if args.headline_only:
            return
        failed, dt_loc, entry, msg = False, 0.0, None, None
        CURRENT_BLOCK[0] = name
        life.arm(name, args.block_timeout)
        try:
            if not rank0_only or rank == 0:
                inject_failure(name, "start")
                dt_loc, entry = body()
        except Exception as exc:
            failed, msg = True, repr(exc)
        CURRENT_BLOCK[0] = None
        try:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        except Exception as exc:
            failed, msg = True, msg or repr(exc)
        try:
            any_failed, dt_max = (failed, dt_loc) if rank0_only else agree(failed, dt_loc)
        except Exception as exc:   # the process group is gone (a rank died): nothing more can be measured
            life.print_line({name: {"error": "the ranks' agreement after the block failed (a rank died?): " + repr(exc)}})
            os._exit(1)
        life.disarm()
        if rank == 0:
            if any_failed:
                result[name] = {"error": msg or "the block failed on another rank"}
            else:
                try:
                    result[name] = entry(dt_max)
                except Exception as exc:
                    result[name] = {"error": repr(exc)}

    def ceilings():   # measured ceilings of this box, into the headline's roofline object
        try:
            result["roofline"]["measured_copy_GBs"] = copy_ceiling(torch, dev)
        except Exception:
            result["roofline"]["measured_copy_GBs"] = None
        try:   # the routine's own mix (3 reads : 1 write, in place) as a linear aligned stream
            plain = M.stream_ceiling(nontemporal=False)
            nt = M.stream_ceiling(nontemporal=True)
            result["roofline"]["measured_stream_3r1w_GBs"] = {"plain": plain, "nontemporal": nt}
            result["roofline"]["frac_of_measured_ceiling"] = result["roofline"]["achieved"] / max(plain, nt)
        except Exception:
            result["roofline"]["measured_stream_3r1w_GBs"] = None

    # the round-2 protocol (ONE plan, the field sets share u, w, serpentine tile order on): what consecutive
    # tracers of one CRM step get
    def b_shared():
        M.set_serpentine(1)
        try:
            ssteps = min(steps, 40)
            dt6, kms6, _ = bench_plan(M, torch, dist, world, dev, shared, sh_f1, n_loc, n_glob, sl0, nx, nz, 1, ssteps,
                                      SIDE_WARMUP, args.dist, npdt, tdt, 0.3 * mem_frac, shared_uw=True)
        finally:
            M.set_serpentine(serp)
        return dt6, lambda dtm: {
            "workload": "the headline workload with u, w SHARED by all timed launches (one plan, the field sets are "
                        "its tracers) and the serpentine tile order on: 1.05 of the 2.14 GB of a launch are the "
                        "previous launch's bytes, part of them still in the Infinity Cache -- NOT a cold call",
            "value": cells_1 * ssteps / dtm, "unit": "cell-updates/s", "steps": ssteps,
            "ms_per_step": dtm / ssteps * 1e3, "roofline": roofline_block(alg_bytes, kms6)}

    # one step on FRESH reference-layout u, w (mpdata_plan_run_uw): the layout entry of the velocities is
    # inside the timed region, every step
    def b_fresh_uw():
        fsteps = min(steps, 40)
        dt7, kms7, nfs = bench_fresh_uw(M, torch, dist, world, dev, shared, sh_f1, n_loc, n_glob, sl0, nx, nz, fsteps,
                                        SIDE_WARMUP, args.dist, npdt, tdt)
        return dt7, lambda dtm: {
            "workload": f"mpdata_plan_run_uw: f resident in the plan layout, u and w handed over as reference-layout "
                        f"device arrays EVERY step (distinct arrays per step, {nfs} sets), ncrms={n_loc}/GPU, 1 tracer",
            "value": cells_1 * fsteps / dtm, "unit": "cell-updates/s", "steps": fsteps,
            "ms_per_step": dtm / fsteps * 1e3, "roofline": roofline_block(alg_bytes, kms7)}

    # the headline protocol at TWICE the instances per GPU (north_star: "at ncrms >= 65 536"): the fixed part
    # of a launch (ramp, drain of the last wave round, gap to the next launch) halves
    def b_x2():
        n2 = 2 * n_loc
        xsteps = min(steps, 16)
        shared2, _, sh2 = make_shared(M, torch, dev, n2, n2 * world, rank * n2, nx, nz, args.dist, tdt)
        dt8, kms8, info8 = bench_plan(M, torch, dist, world, dev, shared2, sh2["f"], n2, n2 * world, rank * n2, nx, nz, 1,
                                      xsteps, SIDE_WARMUP, args.dist, npdt, tdt, 0.5 * mem_frac)
        del shared2
        return dt8, lambda dtm: {
            "workload": f"the headline protocol (cold: a plan of its own per timed step, {info8['field_sets']} sets) at "
                        f"ncrms={n2}/GPU nx={nx} nz={nz} fp64, 1 tracer",
            "value": 2 * cells_1 * xsteps / dtm, "unit": "cell-updates/s", "steps": xsteps,
            "ms_per_step": dtm / xsteps * 1e3,
            "roofline": roofline_block(M.algorithmic_bytes(n2, nx, nz, 1), kms8)}

    # nz > 64 (the plan kernel's window form; 72 levels: the last window of an instance is a share of a tail wave), the
    # headline protocol at about the headline's cell count
    def b_nz72():
        n72, z72 = max(2, (3 * n_loc // 8) & ~1), 72
        xsteps = min(steps, 16)
        shared72, _, sh72 = make_shared(M, torch, dev, n72, n72 * world, rank * n72, nx, z72, args.dist, tdt)
        dt72, kms72, info72 = bench_plan(M, torch, dist, world, dev, shared72, sh72["f"], n72, n72 * world, rank * n72, nx, z72, 1,
                                         xsteps, SIDE_WARMUP, args.dist, npdt, tdt, 0.5 * mem_frac)
        del shared72
        return dt72, lambda dtm: {
            "workload": f"the headline protocol (cold, {info72['field_sets']} sets) at ncrms={n72}/GPU nx={nx} nz={z72} fp64, 1 tracer: "
                        "several waves per instance",
            "value": n72 * nx * (z72 - 1) * xsteps / dtm, "unit": "cell-updates/s", "steps": xsteps,
            "ms_per_step": dtm / xsteps * 1e3,
            "roofline": roofline_block(M.algorithmic_bytes(n72, nx, z72, 1), kms72)}

    # The headline protocol with TWO launches in flight: consecutive steps are independent (a plan of its own each), so a
    # caller that has more than one batch of CRM instances to advect can alternate between two streams -- the drain of one
    # launch (its last wave round ends spread over ~40 us) then overlaps the ramp of the next.  NOT the headline: there a
    # step starts when the previous one has ended, as the reference's timed region is one call (:109-110).  No kernel
    # duration is claimed from this block (two kernels share the chip): throughput only.
    def b_two_in_flight():
        tsteps = min(steps, 40)
        sets = PlanSets(M, torch, dev, shared, sh_f1, n_loc, n_glob, sl0, nx, nz, 1, tsteps, SIDE_WARMUP, args.dist, npdt,
                        tdt, 0.4 * mem_frac)
        try:
            cur = torch.cuda.current_stream()
            streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            for i, pl in enumerate(sets.plans):
                pl.set_stream(streams[i % 2])
            for s_ in streams:
                s_.wait_stream(cur)
            nt = sets.ntimed - sets.ntimed % 2 or sets.ntimed      # an even cycle: the streams alternate strictly
            go = lambda i: sets.plans[i % nt].run()
            wake_up(torch, lambda n: sets.plans[sets.nset - 1 - (n % sets.nscr)].run(), PREWARM_MS, PREWARM_MIN_MS)
            for i in range(SIDE_WARMUP):
                go(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(tsteps):
                go(i)
            torch.cuda.synchronize()
            dt9 = time.perf_counter() - t0
            for pl in sets.plans:
                pl.set_stream(cur)
        finally:
            sets.close(torch)
        return dt9, lambda dtm: {
            "workload": "the headline workload (cold: a plan of its own per step), consecutive steps on two alternating HIP "
                        "streams: two launches in flight -- what a caller with several independent batches gets; NOT the "
                        "headline, and no per-kernel duration is claimed (the two kernels share the chip)",
            "value": cells_1 * tsteps / dtm, "unit": "cell-updates/s", "steps": tsteps, "ms_per_step": dtm / tsteps * 1e3,
            "algorithmic_GBs": alg_bytes * tsteps / dtm / 1e9, "frac_of_8TBs_throughput": alg_bytes * tsteps / dtm / 1e9 / HBM_PEAK_GBS,
            "timing": "wall clock between two device synchronisations (max over the ranks)"}

    # BASELINE configs[3] / [4]: 25 tracers per instance, every rank
    def b_batched():
        bt = args.batched_tracers
        bsteps, bwarm = min(steps, args.batched_steps), min(warmup, 2)
        dt2, kms2, info2 = bench_plan(M, torch, dist, world, dev, shared, sh_f1, n_loc, n_glob, sl0, nx, nz, bt,
                                      bsteps, bwarm, args.dist, npdt, tdt, mem_frac)

        def entry(dtm):
            ab = M.algorithmic_bytes(n_loc, nx, nz, bt, f32=f32)
            ka = kavg(kms2)
            key = f"{args.variant}_ncrms{n_loc}_nx{nx}_nz{nz}_t{bt}" + ("_f32" if f32 else "") + \
                  ("_wm" if info2["layout"].startswith("wave") else "")
            tr2 = traffic_lookup(key)
            rb = roofline_block(ab, kms2, {"traffic": tr2, "traffic_source": None if tr2 is None else
                                           "profiles/hbm_traffic.json[%s] (recorded profile, not this run)" % key})
            rb["hbm_frac"] = rb["frac"]
            vi = None if f32 else valu_instructions_per_launch(args.variant, n_loc, nx, nz, bt)
            if vi:   # second ceiling (SURVEY.md 7 hard part 3): fp64 VALU issue
                t_valu = vi[0] / VALU_PEAK_WAVE_INSTR_PER_S
                rb["valu_frac"] = t_valu / (ka * 1e-3)
                rb["valu_source"] = "recorded profile (%s), NOT a measurement of this run" % vi[1]
                t_hbm = ab / (HBM_PEAK_GBS * 1e9)
                rb["binding_floor_frac"] = max(t_hbm, t_valu) / (ka * 1e-3)
                rb["binding_floor_note"] = "max(algorithmic bytes / 8 TB/s, VALU instructions / measured issue peak) / kernel time"
                rb["valu_note"] = "VALU instructions per launch / measured fp64 VALU issue peak (tools/valu_rate.hip: 33e12 " \
                                  "lane-ops/s at 2.0 GHz; this kernel runs at 1.7-1.85 GHz, power-limited) / kernel time"
            return {
                "workload": f"BASELINE.json configs[{3 if world == 1 else 4}]: ncrms={n_loc}/GPU (global {n_glob}), "
                            f"{bt} tracers sharing u,w,rho,rhow,adz, plan API",
                "value": cells_1 * bt * bsteps / dtm, "unit": "cell-updates/s", "n_gpus": world,
                "steps": bsteps, "ms_per_step": dtm / bsteps * 1e3, "field_sets": info2["field_sets"],
                "steps_per_field_set": info2["steps_per_field_set"], "device_layout": info2["layout"],
                "timing": "max over the ranks of each rank's own K steps (no barrier inside a side block)",
                "roofline": rb}
        return dt2, entry

    # the reference-layout device call (x-march kernel): what a caller gets whose device arrays stay in the
    # reference layout
    def b_reflayout():
        nb = min(steps + N_SCRATCH, 24)
        fs = []
        for b in range(nb):
            f = alloc(sh["f"], "f")
            M.fill_synthetic(f, "f", 100 + b, args.dist, ncrms_global=n_glob, sl0=sl0)
            fs.append(f)

        def launch_ref(i):
            M.advect_scalar2D(fs[i % nb], shared["u"], shared["w"], shared["rho"], shared["rhow"], shared["flux"], shared["adz"])

        rsteps = min(steps, 40)
        dt4, kms4 = timed_loop(torch, dist, world, launch_ref, rsteps, SIDE_WARMUP, prewarm_launch=launch_ref, collective=False)
        del fs

        def entry(dtm):
            key = f"{args.variant}_ncrms{n_loc}_nx{nx}_nz{nz}_t1" + ("_f32" if f32 else "")
            tr4 = traffic_lookup(key)
            return {
                "workload": f"mpdata_advect_scalar2d{'_f32' if f32 else ''}_device on reference-layout device arrays "
                            f"(x-march kernel), ncrms={n_loc}/GPU, 1 tracer",
                "value": cells_1 * rsteps / dtm, "unit": "cell-updates/s", "steps": rsteps,
                "ms_per_step": dtm / rsteps * 1e3,
                "roofline": roofline_block(alg_bytes, kms4, {"traffic": tr4, "traffic_source": None if tr4 is None else
                                                             "profiles/hbm_traffic.json[%s] (recorded profile)" % key})}
        return dt4, entry

    # the same workload in fp32 (reference precision switch)
    def b_fp32():
        sh32, alloc32, _ = make_shared(M, torch, dev, n_loc, n_glob, sl0, nx, nz, args.dist, torch.float32)
        nb = min(steps + N_SCRATCH, 24)
        fs3 = []
        for b in range(nb):
            f = alloc32(sh["f"], "f")
            M.fill_synthetic(f, "f", 100 + b, args.dist, ncrms_global=n_glob, sl0=sl0)
            fs3.append(f)

        def launch32(i):
            M.advect_scalar2D(fs3[i % nb], sh32["u"], sh32["w"], sh32["rho"], sh32["rhow"], sh32["flux"], sh32["adz"])

        s3 = min(steps, 40)
        dt3, kms3 = timed_loop(torch, dist, world, launch32, s3, SIDE_WARMUP, prewarm_launch=launch32, collective=False)
        del fs3
        torch.cuda.empty_cache()
        # ... and through an fp32 plan (wave-major layout, two instances per lane)
        dt5, kms5, info5 = bench_plan(M, torch, dist, world, dev, sh32, sh_f1, n_loc, n_glob, sl0, nx, nz, 1, s3,
                                      SIDE_WARMUP, args.dist, np.float32, torch.float32, 0.3 * mem_frac)
        del sh32

        def entry(dtm):   # (dtm: the plan loop; the reference-layout loop reports this rank's own time)
            ab = M.algorithmic_bytes(n_loc, nx, nz, 1, f32=True)
            return {
                "workload": f"ncrms={n_loc}/GPU nx={nx} nz={nz} fp32, 1 tracer, plan API ({info5['layout']})",
                "value": cells_1 * s3 / dtm, "unit": "cell-updates/s", "steps": s3,
                "ms_per_step": dtm / s3 * 1e3, "roofline": roofline_block(ab, kms5),
                "reference_layout_device_call": {
                    "workload": "mpdata_advect_scalar2d_f32_device on reference-layout device arrays (x-march kernel)",
                    "value": n_loc * nx * (nz - 1) * s3 / dt3, "ms_per_step": dt3 / s3 * 1e3,
                    "note": "rank 0's own shard and time", "roofline": roofline_block(ab, kms3)}}
        return dt5, entry

    # the EXACT variant -- what a caller of the library gets who sets nothing (the library's and the Fortran driver's
    # default: f AND flux bit-identical to the reference): plan run with 1 and with 25 tracers, reference-layout
    # device call; cold like the headline, fewer steps
    def b_exact():
        M.set_variant(M.VARIANT_EXACT)
        try:
            es = min(steps, 20)
            dt1, k1, i1 = bench_plan(M, torch, dist, world, dev, shared, sh_f1, n_loc, n_glob, sl0, nx, nz, 1, es,
                                     SIDE_WARMUP, args.dist, npdt, tdt, 0.4 * mem_frac)
            bt, bs = args.batched_tracers, min(steps, 6)
            dt25, k25, i25 = bench_plan(M, torch, dist, world, dev, shared, sh_f1, n_loc, n_glob, sl0, nx, nz, bt, bs,
                                        2, args.dist, npdt, tdt, 0.6 * mem_frac)
            nb = 12
            fs = []
            for b in range(nb):
                f = alloc(sh["f"], "f")
                M.fill_synthetic(f, "f", 100 + b, args.dist, ncrms_global=n_glob, sl0=sl0)
                fs.append(f)

            def launch_ref(i):
                M.advect_scalar2D(fs[i % nb], shared["u"], shared["w"], shared["rho"], shared["rhow"], shared["flux"], shared["adz"])
            dtr, kr = timed_loop(torch, dist, world, launch_ref, es, SIDE_WARMUP, prewarm_launch=launch_ref, collective=False)
            del fs
        finally:
            M.set_variant(M.VARIANT_FAST if args.variant == "fast" else M.VARIANT_EXACT)

        def entry(dtm):
            return {
                "workload": "the headline protocol in the EXACT variant (-ffp-contract=off, IEEE divisions; f and flux "
                            "bit-identical to the reference: the limited vertical fluxes of a lane are kept in registers "
                            "and added onto the finished upwind sum in the reference's order, :545, :624)",
                "value": cells_1 * es / dtm, "unit": "cell-updates/s", "steps": es, "ms_per_step": dtm / es * 1e3,
                "roofline": roofline_block(alg_bytes, k1),
                "tracer_batched": {"tracers": bt, "steps": bs, "value": n_loc * nx * (nz - 1) * bt * bs / dt25,
                                   "ms_per_step": dt25 / bs * 1e3, "note": "rank 0's own shard and time",
                                   "roofline": roofline_block(M.algorithmic_bytes(n_loc, nx, nz, bt), k25)},
                "reference_layout_device_call": {"value": n_loc * nx * (nz - 1) * es / dtr, "ms_per_step": dtr / es * 1e3,
                                                 "note": "rank 0's own shard and time; the park array of this kernel family "
                                                         "is allocated and freed in stream order around every call",
                                                 "roofline": roofline_block(alg_bytes, kr)}}
        return dt1, entry

    # end to end: the drop-in call on HOST arrays, H2D + kernel + D2H -- never `value`; N = 1 only
    def b_host_call():
        host = {}
        for k in ("f", "u", "w", "rho", "rhow", "adz", "flux"):
            t = alloc(sh[k], k)
            M.fill_synthetic(t, k, 300, args.dist, ncrms_global=n_glob, sl0=sl0)
            host[k] = t.cpu().numpy().T   # Fortran order, the reference's shapes (pageable memory)
            del t
        torch.cuda.empty_cache()
        ts = []
        for _ in range(3):   # (the first call also pays for the library's buffers)
            t0 = time.perf_counter()
            M.advect_scalar2D_host(host["f"], host["u"], host["w"], host["rho"], host["rhow"], host["flux"], host["adz"])
            ts.append(time.perf_counter() - t0)
        del host
        return min(ts[1:]), lambda dtm: {
            "workload": f"mpdata_advect_scalar2d on host arrays (pageable), ncrms={n_loc} nx={nx} nz={nz}, 1 tracer: "
                        "H2D + kernel + D2H, chunked and pipelined (DESIGN.md 5)",
            "seconds_first_call": ts[0], "seconds": min(ts[1:]),
            "value": cells_1 / min(ts[1:]), "unit": "cell-updates/s",
            "note": "PCIe-inclusive; reported beside the device-resident `value`, never as it"}

    # the second / third kernel, rank 0 only
    def b_bwk():
        r = bench_bwk(torch, dev, min(steps, 50), SIDE_WARMUP, world == 1 and not args.no_cpu_baseline)
        return 0.0, lambda dtm: r

    def b_nlk():
        r = bench_nlk(torch, dev, min(steps, 100), SIDE_WARMUP, world == 1 and not args.no_cpu_baseline)
        return 0.0, lambda dtm: r

    # N > 1: scatter / gather of a root-resident problem over RCCL (outside any timed compute region):
    # ncrms = ncrms_per_gpu * N, one tracer.  It HAS collectives inside and guards them itself (every rank
    # agrees that the root could allocate before the first send / recv).
    def b_scatter():
        r = bench_scatter_gather(M, torch, dist, world, rank, dev, n_loc, nx, nz)
        return 0.0, lambda dtm: r

    try:
        if rank == 0 and not args.headline_only:
            ceilings()
        one = ntr == 1
        if not args.no_shared_block and not args.shared_uw and one:
            side("consecutive_tracers_shared_uw", b_shared)
        if not args.no_fresh_uw and one and not f32:
            side("step_with_fresh_uw", b_fresh_uw)
        if not args.no_x2 and one and not f32 and not args.shared_uw:
            side("twice_the_instances", b_x2)
        if not args.no_x2 and one and not f32 and not args.shared_uw:
            side("two_launches_in_flight", b_two_in_flight)
        if not args.no_x2 and one and not f32 and not args.shared_uw and nz <= 64:
            side("levels_above_64", b_nz72)
        if not args.no_batched and one:
            side("tracer_batched", b_batched)
        if not args.no_reflayout and one:
            side("reference_layout_device_call", b_reflayout)
        if not args.no_fp32 and not f32 and one:
            side("fp32", b_fp32)
        if not args.no_exact and not f32 and one and args.variant == "fast":
            side("exact_variant", b_exact)
        if world == 1 and not args.no_host_call and not f32 and one:
            side("end_to_end_host_call", b_host_call, rank0_only=True)
        if not args.no_bwk and not f32 and one:
            side("biharmonic_wk", b_bwk, rank0_only=True)
            side("high_order_flux", b_nlk, rank0_only=True)
        if world > 1 and not args.no_scatter:
            side("scatter_gather", b_scatter)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(nx, nz)
            except Exception as exc:
                result["cpu_baseline"] = {"error": repr(exc)}
    finally:
        # whatever happened above (an exception outside a block's own handler, Ctrl-C): the line goes out
        life.print_line()
    if world > 1:
        life.arm(None, 60.0 if args.block_timeout > 0 else 0.0)   # (the line is out: a rank that hangs here only costs the exit)
        dist.barrier()
        dist.destroy_process_group()
        life.disarm()


def bench_scatter_gather(M, torch, dist, world, rank, dev, n_loc, nx, nz):
    """Root (rank 0) holds the reference-layout arrays of the global problem; every rank receives
    its contiguous ncrms block (pack kernel -> grouped send/recv = ncclSend/ncclRecv over xGMI ->
    contiguous shard), advects it, and the outputs f, flux are gathered back.  Reports seconds and
    GB/s per peer link (every peer has its own xGMI link to the root)."""
    ns = n_loc * world
    names = ("adz", "f", "u", "w", "rho", "rhow", "flux")
    sh = M.shapes(ns, nx, nz)
    full, ok = None, 1
    if rank == 0:
        try:
            full = {k: torch.empty(sh[k], dtype=torch.float64, device=dev) for k in names}
            for k in names:
                M.fill_synthetic(full[k], k, 100, 1)
        except Exception:   # (e.g. out of memory on the root): every rank must skip the exchange
            full, ok = None, 0
            torch.cuda.empty_cache()
    # all ranks agree before the first send/recv: a rank that raised here would leave the others
    # waiting in the exchange for ever
    flag = torch.tensor([ok], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        raise RuntimeError("root could not allocate the global problem for the scatter/gather measurement")
    arg = full if rank == 0 else {k: sh[k][:-1] for k in names}
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    mine = M.scatter_inputs(arg, ns, src=0, device=dev)
    torch.cuda.synchronize(); dist.barrier()
    t_sc = time.perf_counter() - t0
    M.advect_scalar2D(mine["f"], mine["u"], mine["w"], mine["rho"], mine["rhow"], mine["flux"], mine["adz"])
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    M.gather_outputs({"f": mine["f"], "flux": mine["flux"]}, full, ns, dst=0)
    torch.cuda.synchronize(); dist.barrier()
    t_ga = time.perf_counter() - t0
    per_in = sum(8 * n_loc * (full[k].numel() // ns if rank == 0 else 0) for k in names) if rank == 0 else 0
    per_out = sum(8 * n_loc * (full[k].numel() // ns) for k in ("f", "flux")) if rank == 0 else 0
    return {"ncrms_global": ns, "scatter_s": t_sc, "gather_s": t_ga,
            "bytes_per_peer_scatter": per_in, "bytes_per_peer_gather": per_out,
            "scatter_GBs_per_link": per_in / t_sc / 1e9, "gather_GBs_per_link": per_out / t_ga / 1e9,
            "note": "root -> N-1 peers in one grouped send/recv; includes the pack / unpack kernels"}


if __name__ == "__main__":
    main()
