#!/usr/bin/env python3
"""Helper of tests/test_multi_fake_rccl.py: runs in a process of its own with
MPDATA_HIP_LIB = tests/stubs/libmpdata_hip_fakerccl.so (the product's objects linked against the
recording RCCL stand-in) and MPDATA_MULTI_FORCE_RCCL=1, drives multi-GPU plans whose ranks all sit
on device 0 through the library's RCCL branch, and prints one JSON line with what happened."""
import ctypes, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import codesign_kernels_amd as M
from oracle import oracle as O

assert "fakerccl" in M.lib_path(), M.lib_path()
fake = ctypes.CDLL(os.path.join(ROOT, "tests", "stubs", "libfake_rccl.so"))   # the instance the library loaded
fake.fake_rccl_log.restype = ctypes.c_char_p
O.build_lib()
M.set_variant(M.VARIANT_EXACT)


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a.T)).to("cuda:0")


def to_host(t):
    return np.asfortranarray(t.cpu().numpy().T)


def make(ncrms, nx, nz, T):
    base = O.make_inputs(ncrms, nx, nz, seed=9, dist=3)
    if T > 1:
        base["f"] = np.asfortranarray(np.stack([O.make_inputs(ncrms, nx, nz, seed=90 + t, dist=3)["f"] for t in range(T)], axis=-1))
        base["flux"] = np.asfortranarray(np.stack([base["flux"]] * T, axis=-1))
    return base


def single(inp, shape, T):
    p = M.Plan(*shape, T)
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
    p.download(f, fl)
    p.close()
    return f, fl


def groups(log):
    """[(ops...)] per closed group: ops = ("send"|"recv", rank, peer, count)"""
    out, cur = [], None
    for line in log.splitlines():
        w = line.split()
        if w[0] == "group_start":
            cur = []
        elif w[0] in ("send", "recv") and cur is not None:
            kv = dict(x.split("=") for x in w[1:])
            cur.append((w[0], int(kv["rank"]), int(kv["peer"]), int(kv["count"])))
        elif w[0] == "group_end" and cur is not None:
            out.append(cur); cur = None
    return out


res = {"cases": []}
for origin, devices, T, shape in (("host", [0, 0, 0], 2, (100, 9, 12)), ("host", [0, 0], 1, (131, 32, 28)),
                                  ("device", [0, 0, 0], 2, (70, 7, 12)), ("device", [0, 0], 1, (258, 32, 28))):
    fake.fake_rccl_reset()
    if origin == "host":
        os.environ["MPDATA_MULTI_XFER"] = "rccl"    # host arrays through the root + RCCL (not the default for host data)
    else:
        os.environ.pop("MPDATA_MULTI_XFER", None)   # default: arrays on the root GPU travel over RCCL
    inp = make(*shape, T)
    f1, fl1 = single(inp, shape, T)
    p = M.Plan(*shape, T, devices=devices)
    G = len(devices)
    shards = p.shards()
    if origin == "host":
        p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
        p.run(); p.sync()
        f = np.empty_like(inp["f"], order="F"); fl = np.empty_like(inp["flux"], order="F")
        p.download(f, fl)
    else:
        d = {k: to_dev(v) for k, v in inp.items()}
        p.import_device(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["adz"], d["flux"])
        p.run(); p.sync()
        fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
        p.export_device(fo, flo)
        f, fl = to_host(fo), to_host(flo)
    st = p.transfer_stats()
    ranks_seen = p.ranks_seen
    p.close()
    log = fake.fake_rccl_log().decode()
    gs = groups(log)
    nzm = shape[2] - 1
    rows = {"f": (shape[1] + 6) * nzm, "u": (shape[1] + 5) * nzm, "w": (shape[1] + 4) * shape[2], "k": nzm, "kz": shape[2]}
    # expected (rows) per group: 5 shared arrays, then per tracer f and flux (scatter); per tracer f, flux (gather)
    exp_rows = [rows["u"], rows["w"], rows["k"], rows["kz"], rows["k"]] + [rows["f"], rows["kz"]] * T + [rows["f"], rows["kz"]] * T
    ok_groups = len(gs) == len(exp_rows)
    detail = []
    for gi, (ops, r) in enumerate(zip(gs, exp_rows)):
        scatter = gi < 5 + 2 * T
        want = set()
        for g in range(1, G):
            n = r * shards[g][2]
            want.add(("send", 0, g, n) if scatter else ("send", g, 0, n))
            want.add(("recv", g, 0, n) if scatter else ("recv", 0, g, n))
        if set(ops) != want or len(ops) != len(want):
            ok_groups = False
            detail.append((gi, sorted(ops), sorted(want)))
    res["cases"].append({"origin": origin, "devices": devices, "T": T, "shape": shape,
                         "f_equal": bool(np.array_equal(f, f1)), "flux_equal": bool(np.array_equal(fl, fl1)),
                         "transport": st["transport"], "ranks_seen": ranks_seen, "groups": len(gs),
                         "groups_expected": len(exp_rows), "groups_ok": ok_groups, "detail": detail[:3],
                         "fake_errors": fake.fake_rccl_errors(), "open_groups": fake.fake_rccl_open_groups(),
                         "error_lines": [l for l in log.splitlines() if l.startswith("ERROR")][:5],
                         "matched": sum(1 for l in log.splitlines() if l.startswith("matched"))})
print("RESULT " + json.dumps(res))
