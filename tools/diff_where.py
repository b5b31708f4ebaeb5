import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import codesign_kernels_amd as M
from oracle import oracle as O
from util import run_hip
O.build_lib()
ncrms, nx, nz = 32, 32, 28
inp = O.make_inputs(ncrms, nx, nz, seed=100, dist=1)
M.set_variant(0)
f, flux = run_hip(M, inp)
fr, xr = O.advect(inp)
d = (f != fr)
print("mismatch count", d.sum(), "of", d.size)
print("by k (level):", d.sum(axis=(0, 1)))
print("by column:", d.sum(axis=(0, 2)))
print("by sl:", d.sum(axis=(1, 2)))
print("max abs", np.abs(f - fr).max())
