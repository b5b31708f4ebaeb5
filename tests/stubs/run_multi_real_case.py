#!/usr/bin/env python3
"""Helper of tests/test_multi_real_devices.py: ONE case of a multi-GPU plan on DIFFERENT physical
devices, in a process of its own (a transport that hangs or faults takes this process down, not the
test session; the test can then try the same case again under MPDATA_MULTI_SYNC=1 in a fresh one).

    run_multi_real_case.py <case-json>

case = {"devices": [0, 1, ...], "xfer": "rccl" | "p2p" | "direct" | "default",
        "origin": "host" | "device", "shape": [ncrms, nx, nz], "T": ntracers,
        "run_uw": bool, "variant": 0 | 1, "big": bool}

Checks (all bitwise, EXACT variant): the sharded plan's f, flux == a single-GPU plan's on the same
arrays == the oracle on sampled blocks of instances (the instances are independent, reference
:505-637); what the plan reports: its shards, the transport it used, the ranks its RCCL communicator
saw.  Prints one line `RESULT {...}`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
case = json.loads(sys.argv[1])
if case["xfer"] == "default":
    os.environ.pop("MPDATA_MULTI_XFER", None)
else:
    os.environ["MPDATA_MULTI_XFER"] = case["xfer"]

import numpy as np
import torch
import codesign_kernels_amd as M
from oracle import oracle as O

O.build_lib()
M.set_variant(case.get("variant", 0))
devices = case["devices"]
root = "cuda:%d" % devices[0]
ncrms, nx, nz = case["shape"]
T = case["T"]
nzm = nz - 1
res = {"case": case, "sync_each": os.environ.get("MPDATA_MULTI_SYNC") == "1"}


def to_dev(a, dev=root):
    return torch.from_numpy(np.ascontiguousarray(a.T)).to(dev)


def to_host(t):
    return np.asfortranarray(t.cpu().numpy().T)


def sample_blocks(n, width=64, count=6):
    """[(a, b)] blocks of instances: both ends, every shard boundary, some in between"""
    cuts = sorted({0, max(0, n - width)} | {max(0, min(n - width, M.shard_range(n, len(devices), g)[0] - width // 2))
                                             for g in range(len(devices))} |
                  {int(x) for x in np.linspace(0, max(0, n - width), count)})
    return [(a, min(n, a + width)) for a in cuts]


if case.get("big"):
    # arrays generated ON the root device by the library's counter-based law (no host array of the problem's size)
    sh = M.shapes(ncrms, nx, nz, T)
    torch.cuda.set_device(devices[0])
    d = {k: torch.empty(sh[k], dtype=torch.float64, device=root) for k in ("f", "u", "w", "rho", "rhow", "adz", "flux")}
    for k in ("u", "w", "rho", "rhow", "adz", "flux"):
        M.fill_synthetic(d[k] if (k != "flux" or T == 1) else d[k][0], k, 100, 1)
    for t in range(T):
        M.fill_synthetic(d["f"][t] if T > 1 else d["f"], "f", 100 + t, 1)
        if T > 1 and t:
            d["flux"][t].copy_(d["flux"][0])
    torch.cuda.synchronize()
    inp = None
else:
    inp = O.make_inputs(ncrms, nx, nz, seed=9, dist=3)
    if T > 1:
        inp["f"] = np.asfortranarray(np.stack([O.make_inputs(ncrms, nx, nz, seed=90 + t, dist=3)["f"]
                                               for t in range(T)], axis=-1))
        inp["flux"] = np.asfortranarray(np.stack([inp["flux"]] * T, axis=-1))
    d = {k: to_dev(v) for k, v in inp.items()} if (case["origin"] == "device" or case.get("run_uw")) else None

other = None
if case.get("run_uw"):      # the plan is filled with OTHER velocities than the step's (kept alive: transfers are queued)
    o_ = O.make_inputs(ncrms, nx, nz, seed=1234, dist=3)
    other = {"u": to_dev(o_["u"]), "w": to_dev(o_["w"])}


def run_plan(p, multi):
    """-> (f, flux) as reference-layout torch tensors on the root (device origin) or numpy arrays (host origin)"""
    if case["origin"] == "device" or case.get("run_uw"):
        if other is not None:
            p.import_device(d["f"], other["u"], other["w"], d["rho"], d["rhow"], d["adz"], d["flux"])
            p.run_uw(d["u"], d["w"])
        else:
            p.import_device(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["adz"], d["flux"])
            p.run()
        p.sync()
        fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
        p.export_device(fo, flo)
        p.sync()
        torch.cuda.synchronize()
        return fo, flo
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
    p.download(f, fl)
    return f, fl


# ---- the single-GPU plan (on the root device) --------------------------------------------------------------
torch.cuda.set_device(devices[0])
p1 = M.Plan(ncrms, nx, nz, T)
f1, fl1 = run_plan(p1, False)
p1.close()

# ---- the sharded plan ---------------------------------------------------------------------------------------
p = M.Plan(ncrms, nx, nz, T, devices=devices)
res["ngpus"] = p.ngpus
res["shards"] = p.shards()
res["shards_ok"] = [list(s[1:]) for s in res["shards"]] == [list(M.shard_range(ncrms, len(devices), g))
                                                              for g in range(len(devices))] and \
                   [s[0] for s in res["shards"]] == devices
for rep in range(2):       # twice: the buffers every array reuses are reused across transfers too
    f, fl = run_plan(p, True)
    eq = (lambda a, b: bool(torch.equal(a, b))) if torch.is_tensor(f) else (lambda a, b: bool(np.array_equal(a, b)))
    res["f_equal_single_%d" % rep] = eq(f, f1)
    res["flux_equal_single_%d" % rep] = eq(fl, fl1)
st = p.transfer_stats()
res["transport"] = st["transport"]
res["scatter_s"], res["gather_s"] = st["scatter_s"], st["gather_s"]
res["scatter_GBs_per_link"] = st["scatter_bytes_per_peer"] / max(st["scatter_s"], 1e-12) / 1e9
res["gather_GBs_per_link"] = st["gather_bytes_per_peer"] / max(st["gather_s"], 1e-12) / 1e9
res["ranks_seen"] = p.ranks_seen
res["kernel_ms"] = p.last_kernel_ms()
p.close()

# ---- the oracle on sampled blocks ---------------------------------------------------------------------------
ok_f, ok_fl = True, True
for a, b in sample_blocks(ncrms):
    if inp is not None:
        blk = {k: np.asfortranarray(v[a:b]) for k, v in inp.items()}
    else:
        blk = {k: to_host(d[k][..., a:b]) for k in d}
    fo = to_host(f[..., a:b]) if torch.is_tensor(f) else f[a:b]
    flo = to_host(fl[..., a:b]) if torch.is_tensor(fl) else fl[a:b]
    for t in range(T):
        one = dict(blk)
        if T > 1:
            one["f"] = np.asfortranarray(blk["f"][..., t]); one["flux"] = np.asfortranarray(blk["flux"][..., t])
        fr, flr = O.advect(one, nthreads=2)
        ft = fo[..., t] if T > 1 else fo
        flt = flo[..., t] if T > 1 else flo
        if case.get("variant", 0) == 0:
            ok_f &= bool(np.array_equal(ft, fr))
            ok_fl &= bool(np.array_equal(flt[:, :nzm], flr[:, :nzm]))
        else:
            ok_f &= bool(np.abs(ft - fr).max() < 1e-12)
            ok_fl &= bool(np.all(np.abs(flt[:, :nzm] - flr[:, :nzm]) <= 1e-12 * np.maximum(1, np.abs(flr[:, :nzm]))))
        ok_fl &= bool(np.array_equal(flt[:, nzm], one["flux"][:, nzm]))
res["f_equal_oracle"], res["flux_equal_oracle"] = ok_f, ok_fl
res["ok"] = all(res[k] for k in ("shards_ok", "f_equal_single_0", "flux_equal_single_0", "f_equal_single_1",
                                 "flux_equal_single_1", "f_equal_oracle", "flux_equal_oracle"))
print("RESULT " + json.dumps(res), flush=True)
