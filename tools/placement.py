"""Diagnostic: kernel time vs. where the f buffer lies relative to u and w (HBM channel/bank
mapping).  One pool allocation; f is placed at a sweep of byte offsets inside it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import codesign_kernels_amd as M
ncrms, nx, nz = 65536, 32, 28
M.set_variant(M.VARIANT_FAST)
sh = M.shapes(ncrms, nx, nz)
d = {k: torch.empty(sh[k], dtype=torch.float64, device="cuda") for k in ("u", "w", "rho", "rhow", "adz", "flux")}
for k in d:
    M.fill_synthetic(d[k], k, 100, 1)
nf = 1
for s in sh["f"]:
    nf *= s
pool = torch.empty(nf + (1 << 27), dtype=torch.float64, device="cuda")   # f + 1 GiB of slack
print("u %x w %x pool %x" % (d["u"].data_ptr(), d["w"].data_ptr(), pool.data_ptr()))
def run(f, n):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n):
        M.advect_scalar2D(f, d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"])
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n
f0 = pool[:nf].view(sh["f"])
M.fill_synthetic(f0, "f", 100, 1)
run(f0, 100)  # warm up clocks
offs = [int(x) for x in sys.argv[1:]] or [0, 256, 4096, 1 << 16, 1 << 19, 1 << 20, 3 << 19, 1 << 21, 1 << 22, 1 << 23, 1 << 24, 3 << 23, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 1 << 29]
for off in offs:
    e = off // 8
    f = pool[e:e + nf].view(sh["f"])
    f.fill_(0.5)
    t = run(f, 30)
    print("offset %10d B (%8.3f MiB)  addr %x : %.4f ms" % (off, off / 2**20, f.data_ptr(), t))
