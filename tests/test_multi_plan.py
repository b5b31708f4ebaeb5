"""Multi-GPU plans (include/mpdata_hip.h section 3b, csrc/mpdata_multi.hip).

CPU part: the shard arithmetic of the C library against the Python partition.
GPU part (one GPU is enough): the same scatter -> run -> gather code path with ngpus = 1
(RCCL communicator of one rank) and with the problem cut into 2, 3 blocks that all live on
device 0 (transports p2p and direct; RCCL itself refuses two ranks on one device) -- results
must be BITWISE those of a single-GPU plan (sharding is exact: no statement of the reference
routine couples two CRM instances, reference :505-637).
"""
import os

import numpy as np
import pytest


def test_shard_range_matches_python_partition(mpdata):
    from codesign_kernels_amd.shard import partition
    for ncrms in (1, 2, 7, 64, 65, 4096, 65536, 524288, 1000003):
        for ngpus in (1, 2, 3, 4, 8):
            if ncrms < ngpus:
                continue
            blocks = [mpdata.shard_range(ncrms, ngpus, g) for g in range(ngpus)]
            assert blocks == [partition(ncrms, ngpus, g) for g in range(ngpus)]
            assert blocks[0][0] == 0 and sum(n for _, n in blocks) == ncrms
            for (a, n), (b, _) in zip(blocks, blocks[1:]):
                assert a + n == b          # contiguous, in order
            assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1


def _single(M, inp, ntr):
    ncrms, nxp6, nzm = inp["f"].shape[:3]
    p = M.Plan(ncrms, nxp6 - 6, nzm + 1, ntr)
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    p.close()
    return f, flux


def _make(oracle, ncrms, nx, nz, T):
    base = oracle.make_inputs(ncrms, nx, nz, seed=9, dist=3)
    if T > 1:
        base["f"] = np.asfortranarray(np.stack([oracle.make_inputs(ncrms, nx, nz, seed=90 + t, dist=3)["f"]
                                                for t in range(T)], axis=-1))
        base["flux"] = np.asfortranarray(np.stack([base["flux"]] * T, axis=-1))
    return base


@pytest.mark.gpu
@pytest.mark.parametrize("xfer", ["rccl", "p2p", "direct"])
@pytest.mark.parametrize("devices,T,shape", [([0], 1, (70, 32, 28)), ([0, 0], 1, (131, 32, 28)),
                                              ([0, 0, 0], 3, (100, 9, 12)), ([0, 0], 2, (5, 3, 7))])
def test_multi_plan_equals_single_plan_bitwise(mpdata, oracle, monkeypatch, xfer, devices, T, shape):
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    monkeypatch.setenv("MPDATA_MULTI_XFER", xfer)
    inp = _make(oracle, *shape, T)
    f1, fl1 = _single(M, inp, T)
    p = M.Plan(*shape, T, devices=devices)
    assert p.ngpus == len(devices)
    sh = p.shards()
    assert [s[1:] for s in sh] == [M.shard_range(shape[0], len(devices), g) for g in range(len(devices))]
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    st = p.transfer_stats()
    assert p.last_kernel_ms() > 0
    p.close()
    assert np.array_equal(f, f1)
    assert np.array_equal(flux, fl1)
    # a repeated device cannot use RCCL (one rank per device): the library says what it used
    assert st["transport"] == (xfer if (xfer != "rccl" or len(set(devices)) == len(devices)) else "p2p")
    assert st["scatter_s"] > 0 and st["gather_s"] > 0 and st["scatter_bytes_per_peer"] > 0


@pytest.mark.gpu
def test_multi_plan_matches_oracle_and_env_device_list(mpdata, oracle, monkeypatch):
    """mpdata_plan_create_multi (the ngpus form the Fortran driver calls) with
    MPDATA_MULTI_DEVICES mapping both ranks to device 0; checked against the oracle."""
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    monkeypatch.setenv("MPDATA_MULTI_DEVICES", "0,0")
    inp = _make(oracle, 96, 32, 28, 1)
    p = M.Plan(96, 32, 28, 1, ngpus=2)
    assert p.ngpus == 2
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    p.close()
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    assert np.array_equal(f, f_ref)
    assert np.array_equal(flux[:, :-1], flux_ref[:, :-1])     # (EXACT plans: flux bit-identical too)
    assert np.array_equal(flux[:, -1], inp["flux"][:, -1])


@pytest.mark.gpu
def test_multi_plan_argument_errors(mpdata):
    M = mpdata
    with pytest.raises(M.MpdataError):
        M.Plan(64, 32, 28, 1, devices=[0, 99])
    with pytest.raises(M.MpdataError):
        M.Plan(2, 32, 28, 1, devices=[0, 0, 0])     # fewer instances than GPUs


@pytest.mark.gpu
def test_shard_plans_are_ordinary_plans(mpdata, oracle, monkeypatch):
    """mpdata_plan_shard_plan: the per-GPU plan of a multi-GPU plan takes device-side import /
    export of its block (a caller whose shards already live on the devices)."""
    import ctypes
    import torch
    from util import to_dev, to_host
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    monkeypatch.setenv("MPDATA_MULTI_XFER", "direct")
    ncrms, nx, nz = 90, 12, 9
    inp = oracle.make_inputs(ncrms, nx, nz, seed=4, dist=1)
    p = M.Plan(ncrms, nx, nz, 1, devices=[0, 0])
    L = M.lib()
    f_ref, flux_ref = oracle.advect(inp, nthreads=2)
    outs = []
    for g, (dev, sl0, nloc) in enumerate(p.shards()):
        sub = L.mpdata_plan_shard_plan(p._p, g)
        assert sub
        blk = {k: to_dev(np.asfortranarray(v[sl0:sl0 + nloc])) for k, v in inp.items()}
        args = [ctypes.c_void_p(blk[k].data_ptr()) for k in ("f", "u", "w", "rho", "rhow", "adz", "flux")]
        assert L.mpdata_plan_import_device(ctypes.c_void_p(sub), *args, 0, 1) == 0
        outs.append((sub, blk, sl0, nloc))
    p.run(); p.sync()
    for sub, blk, sl0, nloc in outs:
        assert L.mpdata_plan_export_device(ctypes.c_void_p(sub), ctypes.c_void_p(blk["f"].data_ptr()),
                                           ctypes.c_void_p(blk["flux"].data_ptr()), 0, 1) == 0
    p.sync()
    torch.cuda.synchronize()
    for sub, blk, sl0, nloc in outs:
        assert np.array_equal(to_host(blk["f"]), f_ref[sl0:sl0 + nloc])
    assert L.mpdata_plan_shard_plan(p._p, 5) is None
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("devices,T,shape", [([0, 0], 1, (131, 32, 28)), ([0, 0, 0], 2, (100, 9, 12))])
def test_device_origin_scatter_gather_equals_single_plan(mpdata, oracle, monkeypatch, devices, T, shape):
    """Arrays of the GLOBAL problem on the root GPU (mpdata_plan_import_device / _export_device on the
    multi-GPU plan): pack kernel + peer transfers, every array queued behind the previous one without a
    host synchronisation in between.  With every rank on device 0 the transport is the peer-copy one
    (events order the reuse of the packed buffers); the RCCL branch of the same code runs in
    tests/test_multi_fake_rccl.py.  Bitwise equal to a single-GPU plan, twice in a row (buffer reuse)."""
    import torch
    from util import to_dev, to_host
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    monkeypatch.delenv("MPDATA_MULTI_XFER", raising=False)
    inp = _make(oracle, *shape, T)
    f1, fl1 = _single(M, inp, T)
    d = {k: to_dev(v) for k, v in inp.items()}
    p = M.Plan(*shape, T, devices=devices)
    for _ in range(2):
        p.import_device(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["adz"], d["flux"])
        p.run(); p.sync()
        fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
        p.export_device(fo, flo)
        assert np.array_equal(to_host(fo), f1) and np.array_equal(to_host(flo), fl1)
    st = p.transfer_stats()
    assert st["transport"] == "p2p" and st["scatter_bytes_per_peer"] > 0 and st["gather_bytes_per_peer"] > 0
    assert p.ranks_seen == 0      # no RCCL communicator: RCCL refuses two ranks on one device
    p.close()


@pytest.mark.gpu
def test_plan_timing_switch(mpdata, oracle):
    """mpdata_plan_set_timing(0): no event pair around the runs, last_kernel_ms says so; results unchanged."""
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    inp = _make(oracle, 64, 32, 28, 1)
    f1, fl1 = _single(M, inp, 1)
    p = M.Plan(64, 32, 28, 1)
    p.set_timing(False)
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    with pytest.raises(M.MpdataError):
        p.last_kernel_ms()
    f = np.empty_like(inp["f"], order="F"); flux = np.empty_like(inp["flux"], order="F")
    p.download(f, flux)
    assert np.array_equal(f, f1) and np.array_equal(flux, fl1)
    p.set_timing(True)
    p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
    p.run(); p.sync()
    assert p.last_kernel_ms() > 0
    p.close()


@pytest.mark.gpu
def test_multi_plan_run_uw_scatters_fresh_velocities(mpdata, oracle):
    """mpdata_plan_run_uw on a multi-GPU plan: u, w = full-width arrays on the root GPU, scattered, then every
    shard runs; the plan was filled with OTHER velocities.  Equal to the oracle on (f, u, w)."""
    from util import to_dev, to_host
    import torch
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    shape = (150, 32, 28)
    inp = _make(oracle, *shape, 1)
    other = oracle.make_inputs(*shape, seed=1234, dist=3)
    d = {k: to_dev(v) for k, v in inp.items()}
    p = M.Plan(*shape, 1, devices=[0, 0, 0])
    p.import_device(d["f"], to_dev(other["u"]), to_dev(other["w"]), d["rho"], d["rhow"], d["adz"], d["flux"])
    p.run_uw(d["u"], d["w"])
    p.sync()
    fo, flo = torch.empty_like(d["f"]), torch.empty_like(d["flux"])
    p.export_device(fo, flo)
    p.close()
    f_ref, flux_ref = oracle.advect(inp, nthreads=4)
    assert np.array_equal(to_host(fo), f_ref)
    assert np.array_equal(to_host(flo)[:, :-1], flux_ref[:, :-1])


@pytest.mark.gpu
def test_multi_plan_upload_without_flux_zeroes_it_every_time(mpdata, oracle, monkeypatch):
    """An upload with flux = NULL hands every tracer a zero flux array on EVERY upload, as a single-GPU plan
    does -- also when an earlier upload of the same plan carried values (flux(:,nz) is never written by the
    routine, reference :541, :624, so stale values would come back with the download)."""
    M = mpdata
    M.set_variant(M.VARIANT_EXACT)
    monkeypatch.setenv("MPDATA_MULTI_XFER", "direct")
    shape, T = (70, 9, 12), 2
    inp = _make(oracle, *shape, T)
    assert np.abs(inp["flux"][:, -1]).max() > 0
    outs = []
    for devices in (None, [0, 0]):
        p = M.Plan(*shape, T) if devices is None else M.Plan(*shape, T, devices=devices)
        p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], inp["flux"])
        p.upload(inp["f"], inp["u"], inp["w"], inp["rho"], inp["rhow"], inp["adz"], None)
        p.run(0, 1); p.sync()       # tracer 1 is never run: its flux stays what the upload made it
        f = np.empty_like(inp["f"], order="F"); flux = np.full_like(inp["flux"], 7.0, order="F")
        p.download(f, flux)
        p.close()
        assert np.all(flux[:, -1, :] == 0.0) and np.all(flux[..., 1] == 0.0)
        outs.append((f, flux))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.gpu
def test_multi_plan_refuses_full_width_arrays_that_are_not_on_its_root_gpu(mpdata, oracle):
    """import_device / export_device / run_uw of a multi-GPU plan dereference the caller's arrays in a pack
    kernel on the ROOT GPU: a pointer HIP attributes to the host (here: pinned host memory) is refused with
    MPDATA_EINVAL instead of being handed to that kernel; mpdata_plan_device_alloc allocates on the root."""
    import ctypes
    import torch
    from util import to_dev
    M = mpdata
    shape = (64, 9, 12)
    inp = _make(oracle, *shape, 1)
    d = {k: to_dev(v) for k, v in inp.items()}
    p = M.Plan(*shape, 1, devices=[0, 0])
    pinned = torch.empty(d["u"].shape, dtype=torch.float64).pin_memory()
    L = M.lib()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = L.mpdata_plan_import_device(p._p, vp(d["f"]), vp(pinned), vp(d["w"]), vp(d["rho"]), vp(d["rhow"]), vp(d["adz"]),
                                     vp(d["flux"]), 0, 1)
    assert rc == M.EINVAL and b"root GPU" in L.mpdata_last_error()
    p.import_device(d["f"], d["u"], d["w"], d["rho"], d["rhow"], d["adz"], d["flux"])
    assert L.mpdata_plan_run_uw(p._p, 0, 1, vp(pinned), vp(d["w"])) == M.EINVAL
    pf = torch.empty(d["f"].shape, dtype=torch.float64).pin_memory()
    assert L.mpdata_plan_export_device(p._p, vp(pf), None, 0, 1) == M.EINVAL
    ptr = ctypes.c_void_p()
    assert L.mpdata_plan_device_alloc(p._p, ctypes.byref(ptr), 4096) == 0 and ptr.value
    assert L.mpdata_device_free(ptr) == 0
    p.close()
