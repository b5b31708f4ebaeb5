"""Diagnostic: kernel time of biharmonic_wk_scalar (nelemd=5400) for library builds given on the
command line (paths or '-' for the default), both variants."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "-":
            env["BWK_HIP_LIB"] = os.path.abspath(lib)
        subprocess.run([sys.executable, __file__, "--child", lib], env=env, check=True)
    sys.exit(0)
import torch
import codesign_kernels_amd.bwk as K
nelemd, nlev, qsize = 5400, 72, 40
g = torch.Generator(device="cuda").manual_seed(11)
q = torch.rand((nelemd, qsize, nlev, 4, 4), dtype=torch.float64, device="cuda", generator=g)
el = torch.rand((nelemd, 144), dtype=torch.float64, device="cuda", generator=g)
dv = torch.rand((4, 4), dtype=torch.float64, device="cuda", generator=g)
for var, name in ((K.VARIANT_FAST, "fast"), (K.VARIANT_EXACT, "exact")):
    K.set_variant(var)
    for _ in range(60):
        K.biharmonic_wk_scalar(el, q, dv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60):
        K.biharmonic_wk_scalar(el, q, dv)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 60
    print("%-40s %-5s %.4f ms  %.2f TB/s" % (sys.argv[2], name, ms, K.algorithmic_bytes(nelemd, nlev, qsize) / ms / 1e9))
