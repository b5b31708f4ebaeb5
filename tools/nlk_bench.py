#!/usr/bin/env python3
"""Kernel time of the high-order flux nest (libnlk_hip.so, FAST) on the 32 x mesh (819200 edges, 89600
cells, 100 levels, 10 cells per edge): N launches with LOCAL connectivity (cells of an edge within
+-128 cells of its position), then N with the reference's RANDOM connectivity (nested.F90:84-90).
tools/profile_round.sh runs it under rocprofv3 (the summary splits the dispatches in that order).
NLK_HIP_LIB selects an experiment build.  usage: python tools/nlk_bench.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import codesign_kernels_amd.nlk as K

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
K.set_variant(K.VARIANT_FAST)
coef = float(np.float32(2.14))
nE, nC, nV, nA = 25600 * 32, 2800 * 32, 100, 10


def mesh(window):
    g = torch.Generator(device=dev).manual_seed(3)
    rnd = lambda *shape: torch.rand(shape, dtype=torch.float64, device=dev, generator=g)
    if window:
        c0 = (torch.arange(nE, device=dev, dtype=torch.int64) * nC // nE).view(nE, 1)
        off = torch.randint(-window, window + 1, (nE, nA), device=dev, generator=g)
        cells = (torch.clamp(c0 + off, 0, nC - 1) + 1).to(torch.int32)
    else:
        cells = torch.randint(1, nC + 1, (nE, nA), dtype=torch.int32, device=dev, generator=g)
    return {"nAdvCellsForEdge": torch.full((nE,), nA, dtype=torch.int32, device=dev), "advCellsForEdge": cells,
            "minLevelCell": torch.ones((nC,), dtype=torch.int32, device=dev),
            "maxLevelCell": torch.clamp((rnd(nC) * nV * 2).round().to(torch.int32), 3, nV),
            "tracerCur": 15.0 * rnd(nC, nV), "normalThicknessFlux": 15.0 * (0.5 - rnd(nE, nV)),
            "advMaskHighOrder": torch.ones((nE, nV), dtype=torch.float64, device=dev),
            "advCoefs": 20.0 * rnd(nE, nA), "advCoefs3rd": 21.0 * rnd(nE, nA)}


ab = K.algorithmic_bytes(nE, nC, nV, nV, nA)
for name, window in (("local +-128", 128), ("random", 0)):
    d = mesh(window)
    out = torch.zeros((nE, nV), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K.high_order_flux(d, nV, coef, out)   # (counted by the profiler as well: N + 1 dispatches per mesh -- see below)
    e0.record()
    for _ in range(N - 1):
        K.high_order_flux(d, nV, coef, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / max(N - 1, 1)
    print(f"{name:12s}: {ms:.4f} ms  compulsory {ab / ms / 1e6:.0f} GB/s  frac {ab / ms / 1e6 / 8000:.3f}  "
          f"incl. gathers {(ab + nE * nA * nV * 8) / ms / 1e6:.0f} GB/s")
    del d, out
