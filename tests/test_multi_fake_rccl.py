"""The RCCL branch of the multi-GPU plans (csrc/mpdata_multi.hip: ncclCommInitAll, one
ncclGroupStart / ncclGroupEnd of ncclSend / ncclRecv pairs per array, one host thread) executed with
2-3 ranks on ONE device.  RCCL itself refuses two ranks per device, so the product's OBJECTS are
linked, for this test only, against a recording stand-in of the nine RCCL entry points
(tests/stubs/fake_rccl.cpp -> tests/stubs/libmpdata_hip_fakerccl.so, csrc/Makefile target
`fakerccl`); the stand-in turns each matched send / recv pair into an ordered device copy and flags
what would hang real RCCL (an operation outside a group, an unmatched or mis-sized pair, a group left
open).  Checked: results bitwise those of a single-GPU plan for host-origin (MPDATA_MULTI_XFER=rccl)
and root-GPU-origin arrays, and per group exactly one send + one recv per peer with count = rows x
nloc[peer] on the right ranks.  It does not replace a run on 8 GPUs (SCALE), it removes the "never
executed" from the branch."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUBS = os.path.join(ROOT, "tests", "stubs")
FAKE_LIB = os.path.join(STUBS, "libmpdata_hip_fakerccl.so")


def test_product_library_links_rccl_not_the_stand_in():
    """the stand-in lives in the test tree only: the library the package loads names librccl, never libfake_rccl"""
    lib = os.path.join(ROOT, "codesign-kernels_amd", "libmpdata_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built here")
    out = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True, check=True).stdout
    assert "librccl" in out and "fake_rccl" not in out
    import codesign_kernels_amd as M
    assert "fakerccl" not in M.lib_path()


@pytest.mark.gpu
@pytest.mark.parametrize("sync_each", [False, True], ids=["queued", "sync-after-every-array"])
def test_rccl_branch_with_several_ranks_on_one_device(sync_each):
    """sync_each: MPDATA_MULTI_SYNC=1, the bring-up fallback that synchronises every stream after every array of
    a transfer instead of once per transfer (same groups, same results)."""
    if not os.path.exists(FAKE_LIB):
        import shutil
        if not shutil.which("hipcc"):
            pytest.skip("tests/stubs/libmpdata_hip_fakerccl.so not built and no hipcc here")
        subprocess.run(["make", "-C", os.path.join(ROOT, "codesign-kernels_amd", "csrc"), "fakerccl"], check=True,
                       stdout=subprocess.DEVNULL)
    env = dict(os.environ, MPDATA_HIP_LIB=FAKE_LIB, MPDATA_MULTI_FORCE_RCCL="1")
    env.pop("MPDATA_MULTI_XFER", None)
    env.pop("MPDATA_MULTI_SYNC", None)
    if sync_each:
        env["MPDATA_MULTI_SYNC"] = "1"
    r = subprocess.run([sys.executable, os.path.join(STUBS, "run_fake_rccl_case.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert len(res["cases"]) == 4
    for c in res["cases"]:
        assert c["f_equal"] and c["flux_equal"], c
        assert c["transport"] == "rccl", c
        assert c["ranks_seen"] == len(c["devices"]), c
        assert c["fake_errors"] == 0 and c["open_groups"] == 0 and not c["error_lines"], c
        assert c["groups"] == c["groups_expected"] and c["groups_ok"], c
        assert c["matched"] == c["groups"] * (len(c["devices"]) - 1), c
