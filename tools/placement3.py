"""Diagnostic: kernel time vs. the alignment of f, u and w modulo the HBM channel interleave
(each array placed at 2-MiB alignment + its own small offset)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import codesign_kernels_amd as M
ncrms, nx, nz = 65536, 32, 28
M.set_variant(M.VARIANT_FAST)
sh = M.shapes(ncrms, nx, nz)
d = {k: torch.empty(sh[k], dtype=torch.float64, device="cuda") for k in ("rho", "rhow", "adz", "flux")}
for k in d:
    M.fill_synthetic(d[k], k, 100, 1)
def numel(s):
    n = 1
    for x in s: n *= x
    return n
n = {k: numel(sh[k]) for k in ("f", "u", "w")}
SLOT = ((max(n.values()) * 8 + (1 << 21)) >> 21 << 21) + (1 << 21)   # bytes, multiple of 2 MiB, > array + 2 MiB
pool = torch.empty(3 * SLOT // 8 + (1 << 18), dtype=torch.float64, device="cuda")
base = (-pool.data_ptr()) % (1 << 21)   # bytes to the next 2-MiB boundary
def view(slot, off, k):
    e = (base + slot * SLOT + off) // 8
    return pool[e:e + n[k]].view(sh[k])
def run(f, u, w, nrep):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(nrep):
        M.advect_scalar2D(f, u, w, d["rho"], d["rhow"], d["flux"], d["adz"])
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / nrep
combos = [(0, 0, 0), (256, 0, 0), (0, 256, 0), (0, 0, 256), (256, 512, 0), (512, 256, 0), (256, 512, 768), (384, 640, 0),
          (128, 256, 0), (256, 0, 512), (4096 + 256, 8192 + 512, 0), (1024, 2048, 0), (0, 0, 0), (256, 512, 0)]
first = True
for of, ou, ow in combos:
    f, u, w = view(0, of, "f"), view(1, ou, "u"), view(2, ow, "w")
    M.fill_synthetic(u, "u", 100, 1); M.fill_synthetic(w, "w", 100, 1)
    f.fill_(0.5)
    if first:
        run(f, u, w, 150); first = False
    t = run(f, u, w, 40)
    print("f+%5d u+%5d w+%5d : %.4f ms  %.1f Gcu/s" % (of, ou, ow, t, ncrms * nx * (nz - 1) / t / 1e6))
