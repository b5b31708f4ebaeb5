#!/usr/bin/env python3
"""One-off wider sweep than the test suite over round 5's kernel changes (GPU): random shapes through mpdata_plan_run
with nz 3 .. 238 (every lane mapping; above 64 levels several waves per instance), nx 1 .. 40, 1 .. 6 tracers (even and
odd: the one-launch batch with an odd tracer), ncrms 1 .. 700, the three input laws:
  * EXACT: f AND flux bit-identical to the oracle (the register park at nx <= 36, the park array beyond);
  * FAST: within 1e-12 on conditioned inputs, rel-L1 < 1e-14 otherwise;
  * fp32 plans (even ncrms): EXACT f bitwise against the fp32 oracle;
  * the reference-layout device call (x-march kernels; nz > 64: the staged call through a per-thread plan): EXACT bitwise.
What the sweep looks for: lanes switched off for the march (EXEC mask) on every (LPS, nz, chunk size) combination, the
seams of the nz > 64 windows, register-park indices at every nx mod 6.
usage: python tools/fuzz_round5.py [N]      (FUZZ_SEED draws another sweep)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import codesign_kernels_amd as M
from oracle import oracle as O
from util import run_hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
O.build_lib()
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "5151")))
bad = 0


def rshape(nzmax):
    ncrms = int(rng.integers(1, 700))
    nz = int(rng.choice([rng.integers(3, 9), rng.integers(9, 17), rng.integers(17, 33), rng.integers(33, 65),
                         rng.integers(65, nzmax + 1) if nzmax > 64 else rng.integers(33, 65)]))
    nx = int(rng.integers(1, 41))
    if nz > 64:
        ncrms = min(ncrms, 200)
        if rng.integers(0, 4) == 0:
            nx = int(rng.integers(67, 100))      # (EXACT: beyond both register parks -> the park array of the window form)
    return ncrms, nx, nz


def plan_run(inp, shape, T, dtype=np.float64):
    p = M.Plan(*shape, T, dtype=dtype)
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
    p.download(f, fl)
    lay = p.layout
    p.close()
    return f, fl, lay


for it in range(n):
    shape = rshape(238)
    T = int(rng.integers(1, 7)) if it % 3 == 0 else 1
    dist = int(rng.integers(1, 4))
    variant = M.VARIANT_EXACT if it % 2 == 0 else M.VARIANT_FAST
    if variant == M.VARIANT_FAST and dist != 1:
        dist = 1 if it % 4 == 1 else dist
    M.set_variant(variant)
    base = O.make_inputs(*shape, seed=7000 + it, dist=dist)
    fs = [O.make_inputs(*shape, seed=8000 + 10 * it + t, dist=dist)["f"] for t in range(T)]
    inp = dict(base)
    if T > 1:
        inp["f"] = np.asfortranarray(np.stack(fs, axis=-1))
        inp["flux"] = np.asfortranarray(np.stack([base["flux"]] * T, axis=-1))
    else:
        inp["f"] = fs[0]
    f, fl, lay = plan_run(inp, shape, T)
    nzm = shape[2] - 1
    ok = True
    for t in range(T):
        fr, flr = O.advect(dict(base, f=fs[t].copy()), nthreads=4)
        ft, flt = (f[..., t], fl[..., t]) if T > 1 else (f, fl)
        if variant == M.VARIANT_EXACT:
            ok &= bool(np.array_equal(ft, fr)) and bool(np.array_equal(flt, flr))
        elif dist == 1:
            ok &= float(np.abs(ft - fr).max()) < 1e-12 and float(np.abs(flt - flr).max()) < 1e-12
        else:
            ok &= O.rel_l1(ft, fr) < 1e-14 and O.rel_l1(flt[:, :nzm], flr[:, :nzm]) < 1e-14
    if not ok:
        bad += 1
        print("BAD plan", shape, "T", T, "dist", dist, "variant", variant, "layout", lay, flush=True)
    # reference-layout device call (x-march; nz > 64: through the calling thread's plan), EXACT, one tracer
    if it % 4 == 0:
        M.set_variant(M.VARIANT_EXACT)
        one = dict(base, f=fs[0].copy())
        fd, fld = run_hip(M, one)
        fr, flr = O.advect(one, nthreads=4)
        if not (np.array_equal(fd, fr) and np.array_equal(fld, flr)):
            bad += 1
            print("BAD device call", shape, "dist", dist, flush=True)
    # fp32 plan, EXACT f bitwise
    if shape[0] % 2 == 0 and it % 5 == 0:
        M.set_variant(M.VARIANT_EXACT)
        i32 = O.make_inputs(*shape, seed=7000 + it, dist=1, dtype=np.float32)
        fr, flr = O.advect(i32)
        f32, fl32, lay32 = plan_run(i32, shape, 1, dtype=np.float32)
        if not np.array_equal(f32, fr):
            bad += 1
            print("BAD fp32 plan", shape, "layout", lay32, flush=True)
    if (it + 1) % 25 == 0:
        print(f"{it + 1} / {n} cases, {bad} bad", flush=True)
# ---- second part: tracer subsets (mpdata_plan_run_tracers), the serpentine tile order, run_uw on fresh velocities and
#      sharded plans (blocks on one device) over the same shape space, EXACT bitwise
M.set_variant(M.VARIANT_EXACT)
from util import to_dev, to_host
for it in range(n // 2):
    shape = rshape(238)
    T = int(rng.integers(2, 6))
    base = O.make_inputs(*shape, seed=17000 + it, dist=3)
    fs = [O.make_inputs(*shape, seed=18000 + 10 * it + t, dist=3)["f"] for t in range(T)]
    inp = dict(base, f=np.asfortranarray(np.stack(fs, axis=-1)), flux=np.asfortranarray(np.stack([base["flux"]] * T, axis=-1)))
    first = int(rng.integers(0, T)); count = int(rng.integers(1, T - first + 1))
    serp = bool(rng.integers(0, 2))
    M.set_serpentine(1 if serp else 0)
    kind = it % 3
    ok = True
    try:
        if kind == 0:      # subset of the tracers, twice (serpentine: the second run walks the tiles from the other end)
            p = M.Plan(*shape, T)
            p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
            p.run(first, count); p.sync()
            f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
            p.download(f, fl); p.close()
            for t in range(T):
                if first <= t < first + count:
                    fr, flr = O.advect(dict(base, f=fs[t].copy()), nthreads=4)
                    ok &= bool(np.array_equal(f[..., t], fr)) and bool(np.array_equal(fl[..., t], flr))
                else:
                    ok &= bool(np.array_equal(f[..., t], fs[t]))
        elif kind == 1:    # a step on fresh reference-layout velocities (the plan holds OTHER ones)
            other = O.make_inputs(*shape, seed=19000 + it, dist=3)
            d = {k: to_dev(v) for k, v in inp.items()}
            ou, ow = to_dev(other["u"]), to_dev(other["w"])
            p = M.Plan(*shape, T)
            p.import_device(d["f"], ou, ow, d["rho"], d["rhow"], d["adz"], d["flux"])
            p.run_uw(d["u"], d["w"]); p.sync()
            fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
            p.export_device(fo, flo); p.sync(); torch.cuda.synchronize(); p.close()
            f, fl = to_host(fo), to_host(flo)
            for t in range(T):
                fr, flr = O.advect(dict(base, f=fs[t].copy()), nthreads=4)
                ok &= bool(np.array_equal(f[..., t], fr)) and bool(np.array_equal(fl[..., t], flr))
        else:              # the problem cut into 2-3 blocks on one device
            if shape[0] >= 3:
                os.environ["MPDATA_MULTI_XFER"] = "direct" if it % 2 else "p2p"
                p = M.Plan(*shape, T, devices=[0] * int(rng.integers(2, 4)))
                p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
                p.run(); p.sync()
                f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
                p.download(f, fl); p.close()
                for t in range(T):
                    fr, flr = O.advect(dict(base, f=fs[t].copy()), nthreads=4)
                    ok &= bool(np.array_equal(f[..., t], fr)) and bool(np.array_equal(fl[..., t], flr))
    except M.MpdataError as exc:
        ok = False
        print("ERROR", exc, flush=True)
    if not ok:
        bad += 1
        print("BAD part 2", shape, "T", T, "kind", kind, "first", first, "count", count, "serpentine", serp, flush=True)
    if (it + 1) % 25 == 0:
        print(f"part 2: {it + 1} / {n // 2} cases, {bad} bad", flush=True)
M.set_serpentine(0)
print(f"fuzz_round5: {n} + {n // 2} cases, {bad} bad")
sys.exit(1 if bad else 0)
