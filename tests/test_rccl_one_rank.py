"""The torch.distributed calls bench.py makes for N > 1, against REAL RCCL with the one rank a one-GPU box allows:
process-group creation the way bench.py does it (backend "nccl" with device_id), barrier, the all-reduce of the ranks'
agreement (MAX over a device tensor), the all-gather of the device ids, the shard scatter / gather helpers (no peer
with one rank: the root's own block through the pack / unpack kernels), destroy.  What it cannot show is a second
rank; what it does show is that none of these calls is refused by this torch / RCCL build."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    import torch.distributed as dist
    import codesign_kernels_amd as M
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dist.barrier()
    t = torch.tensor([1.0, 0.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.tolist() == [1.0, 0.25]
    flag = torch.tensor([1], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dv = torch.tensor([torch.cuda.current_device()], dtype=torch.int32, device=dev)
    allv = [torch.zeros_like(dv)]
    dist.all_gather(allv, dv)
    assert int(allv[0].item()) == local
    ns, nx, nz = 64, 8, 6
    sh = M.shapes(ns, nx, nz)
    names = ("adz", "f", "u", "w", "rho", "rhow", "flux")
    full = {k: torch.empty(sh[k], dtype=torch.float64, device=dev) for k in names}
    for k in names:
        M.fill_synthetic(full[k], k, 100, 1)
    mine = M.scatter_inputs(full, ns, src=0, device=dev)
    assert all(torch.equal(mine[k], full[k]) for k in names)
    out = {k: torch.zeros_like(full[k]) for k in ("f", "flux")}
    M.gather_outputs({"f": mine["f"], "flux": mine["flux"]}, out, ns, dst=0)
    assert torch.equal(out["f"], full["f"]) and torch.equal(out["flux"], full["flux"])
    torch.cuda.synchronize(); dist.barrier()
    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK")
""") % ROOT


@pytest.mark.gpu
def test_bench_collectives_against_real_rccl_with_one_rank(tmp_path):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "one_rank.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0 and "RCCL_ONE_RANK_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
