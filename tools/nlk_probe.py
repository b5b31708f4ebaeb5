#!/usr/bin/env python3
"""Where does the high-order flux nest spend its time on the 32 x mesh?  Same kernel, inputs that switch
parts of the work off:
  streams only    nAdvCellsForEdge = 0: the three edge streams (2 reads, 1 write) and nothing else
  one cell        every edge gathers cell 1 ten times: the gather instructions, all hits in L1
  local / random  the full nest, cells within +-128 of the edge's position / anywhere in the mesh
  short columns   local connectivity, maxLevelCell = 3 everywhere: gathers issued, 3 of 100 lanes active
usage: python tools/nlk_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import codesign_kernels_amd.nlk as K

dev = torch.device("cuda", 0)
K.set_variant(K.VARIANT_FAST)
coef = float(np.float32(2.14))
nE, nC, nV, nA = 25600 * 32, 2800 * 32, 100, 10
g = torch.Generator(device=dev).manual_seed(3)
rnd = lambda *shape: torch.rand(shape, dtype=torch.float64, device=dev, generator=g)
c0 = (torch.arange(nE, device=dev, dtype=torch.int64) * nC // nE).view(nE, 1)
local = (torch.clamp(c0 + torch.randint(-128, 129, (nE, nA), device=dev, generator=g), 0, nC - 1) + 1).to(torch.int32)
rand = torch.randint(1, nC + 1, (nE, nA), dtype=torch.int32, device=dev, generator=g)
base = {"nAdvCellsForEdge": torch.full((nE,), nA, dtype=torch.int32, device=dev), "advCellsForEdge": local,
        "minLevelCell": torch.ones((nC,), dtype=torch.int32, device=dev),
        "maxLevelCell": torch.clamp((rnd(nC) * nV * 2).round().to(torch.int32), 3, nV),
        "tracerCur": 15.0 * rnd(nC, nV), "normalThicknessFlux": 15.0 * (0.5 - rnd(nE, nV)),
        "advMaskHighOrder": torch.ones((nE, nV), dtype=torch.float64, device=dev),
        "advCoefs": 20.0 * rnd(nE, nA), "advCoefs3rd": 21.0 * rnd(nE, nA)}
out = torch.zeros((nE, nV), dtype=torch.float64, device=dev)
cases = [("streams only", dict(base, nAdvCellsForEdge=torch.zeros((nE,), dtype=torch.int32, device=dev))),
         ("one cell", dict(base, advCellsForEdge=torch.ones((nE, nA), dtype=torch.int32, device=dev))),
         ("local", base), ("random", dict(base, advCellsForEdge=rand)),
         ("short columns", dict(base, maxLevelCell=torch.full((nC,), 3, dtype=torch.int32, device=dev)))]
for mode in (2, 1, 0):
    K.set_kernel(mode)
    for name, d in cases:
        for _ in range(3):
            K.high_order_flux(d, nV, coef, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.high_order_flux(d, nV, coef, out)
        e1.record()
        torch.cuda.synchronize()
        print(f"kernel mode {mode:2d}  {name:14s}: {e0.elapsed_time(e1) / 10:.4f} ms")
