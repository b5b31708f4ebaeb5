"""Diagnostic: per-step kernel times of bench.py's timed loop (clock / power transient)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import codesign_kernels_amd as M
import torch.distributed as dist
steps, warm = int(sys.argv[1]), int(sys.argv[2])
M.set_variant(M.VARIANT_FAST)
dev = torch.device("cuda", 0)
d, fs = bench.make_problem(M, torch, dev, 65536, 65536, 0, 32, 28, 1, steps + warm, 1)
dt, kms = bench.timed_run(M, torch, dist, 1, d, fs, steps, warm)
print("total %.4f ms/step" % (dt / steps * 1e3))
for i in range(0, steps, 10):
    print(i, " ".join("%.3f" % x for x in kms[i:i + 10]))
