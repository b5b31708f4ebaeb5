"""Diagnostic: run the -DMPD2_STAMPS build once and print where a step spends its cycles."""
import os, sys, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import codesign_kernels_amd as M
ncrms, nx, nz = 65536, 32, 28
M.set_variant(M.VARIANT_FAST if (len(sys.argv) > 1 and sys.argv[1] == "fast") else M.VARIANT_EXACT)
sh = M.shapes(ncrms, nx, nz)
d = {k: torch.empty(s, dtype=torch.float64, device="cuda") for k, s in sh.items()}
for k in d:
    M.fill_synthetic(d[k], k, 100, 1)
nblk = ncrms // 16
dbg = torch.zeros(nblk * 256, dtype=torch.int64, device="cuda")
M.lib().mpdata_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
for it in range(3):
    f = d["f"].clone()
    M.advect_scalar2D(f, d["u"], d["w"], d["rho"], d["rhow"], d["flux"], d["adz"])
torch.cuda.synchronize()
M.lib().mpdata_set_debug_buffer(None)
s = dbg.cpu().numpy().reshape(nblk, 256)
t0, r0, t1, r1 = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
clk = (t1 - t0) / ((r1 - r0) / 100e6) / 1e9
dur = (r1 - r0) / 100e6 * 1e6
print("blocks", nblk, "per-WG lifetime us: median %.1f min %.1f max %.1f" % (np.median(dur), dur.min(), dur.max()))
print("shader clock GHz: median %.3f min %.3f max %.3f" % (np.median(clk), clk.min(), clk.max()))
print("kernel span us (first start to last end): %.1f" % ((r1.max() - r0.min()) / 100e6 * 1e6))
NS = nx + 6  # column steps q = -2 .. nx+3
st = s[:, 4:4 + 4 * NS].reshape(nblk, NS, 4).astype(np.float64)
# per step: [before wait, after wait, after barrier, before stage B]
w = st[:, :, 1] - st[:, :, 0]
b = st[:, :, 2] - st[:, :, 1]
a = st[:, :, 3] - st[:, :, 2]
rest = st[:, 1:, 0] - st[:, :-1, 3]
tot = st[:, 1:, 0] - st[:, :-1, 0]
for name, x in (("vmcnt/lgkm wait", w), ("barrier", b), ("DMA+read+stage A", a), ("stage B..D", rest)):
    print("%-18s median per step %7.0f cyc   (steps 10..30: %7.0f)" % (name, np.median(x), np.median(x[:, 10:30])))
print("step total (steady) median %.0f cycles" % np.median(tot[:, 10:30]))
print("median step duration by step index (cycles):")
print(" ".join("%d" % x for x in np.median(tot, axis=0)))
print("vmcnt wait by step:", " ".join("%d" % x for x in np.median(w, axis=0)))
print("barrier by step:   ", " ".join("%d" % x for x in np.median(b, axis=0)))
print("stage A by step:   ", " ".join("%d" % x for x in np.median(a, axis=0)))
print("stage B-D by step: ", " ".join("%d" % x for x in np.median(rest, axis=0)))
r0s = (s[:, 1] - s[:, 1].min()) / 100.0
print("WG start times us: p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(r0s, [10, 50, 90, 100])))
# occupancy of the 512 workgroup slots (256 CUs x 2) over the kernel's life
t_s = (s[:, 1] - s[:, 1].min()) / 100.0
t_e = (s[:, 3] - s[:, 1].min()) / 100.0
edges = np.arange(0, t_e.max() + 20, 20.0)
occ = [int(((t_s < b) & (t_e > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
print("workgroups alive per 20-us window:", " ".join(str(x) for x in occ))
print("sum of lifetimes / (span x 512 slots) = %.3f" % ((t_e - t_s).sum() / (t_e.max() * 512)))
order = np.argsort(t_s)
life = (t_e - t_s)[order]
print("lifetime by start order (mean of each 512): ", " ".join("%.1f" % life[i:i + 512].mean() for i in range(0, nblk, 512)))
