"""The N > 1 control flow of bench.py (barriers, max-over-ranks time, rank-0 JSON line) rehearsed
with two processes.  RCCL refuses two ranks on one device, so the rehearsal mode puts both ranks
on cuda:0 over gloo (MPDATA_BENCH_REHEARSAL=1); the numbers mean nothing, the plumbing is the
same code the driver runs with --gpus N.

The command is the PLAIN one -- `python3 bench.py --gpus 2 ...`, no launcher in the test: bench.py
starts its ranks itself (bench.self_launch: a child `torch.distributed.run`, before torch is imported
and before anything has touched the GPU).  One test keeps the other way in (the ranks started by
torch.distributed.run around bench.py, WORLD_SIZE set)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _two_ranks(extra_env=None, extra_args=(), rc0=True, launcher=False):
    env = dict(os.environ, MPDATA_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):      # (the plain command: nothing has started ranks for it)
        env.pop(k, None)
    cmd = [sys.executable]
    if launcher:
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:   # a port the OS hands out
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"),
            "--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm-ms", "0", "--ncrms-per-gpu", "4096",
            "--batched-tracers", "3", "--batched-steps", "2", "--no-fp32", "--no-bwk", "--no-exact", "--no-cpu-baseline", *extra_args]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert (res.returncode == 0) == rc0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout                     # rank 0 only
    return json.loads(lines[0]), res


@pytest.mark.gpu
def test_plain_command_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` as the driver types it: ONE line, two ranks seen, rc 0."""
    d, res = _two_ranks(extra_args=("--headline-only",))
    assert "BENCH_SELF_LAUNCH " in res.stderr and "torch.distributed.run" in res.stderr
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and len(d["config"]["devices_seen"]) == 2
    assert "bench.py itself" in d["config"]["launched_by"] and d["value"] > 0


@pytest.mark.gpu
def test_ranks_started_by_the_launcher_around_bench():
    d, res = _two_ranks(extra_args=("--headline-only",), launcher=True)
    assert "BENCH_SELF_LAUNCH" not in res.stderr
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["launched_by"] == "torch.distributed.run"


@pytest.mark.gpu
def test_two_rank_bench_prints_one_whole_job_line():
    d, res = _two_ranks()
    # the headline is on stderr the moment it exists, before any side block runs
    early = [ln for ln in res.stderr.splitlines() if "BENCH_HEADLINE {" in ln]
    assert len(early) == 1 and json.loads(early[0].split("BENCH_HEADLINE ", 1)[1])["value"] == d["value"]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["ncrms_global"] == 2 * 4096
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    # configs[4] block (tracer batch on every rank) and the reference-layout side block ran too
    tb = d["tracer_batched"]
    assert tb["n_gpus"] == 2 and tb["steps"] == 2 and tb["value"] > 0 and "configs[4]" in tb["workload"]
    assert d["reference_layout_device_call"]["value"] > 0
    assert d["twice_the_instances"]["value"] > 0 and "ncrms=8192/GPU" in d["twice_the_instances"]["workload"]
    assert "scatter_gather" in d      # (gloo cannot move device tensors: an error entry in the rehearsal)


@pytest.mark.gpu
def test_a_side_block_that_raises_on_one_rank_cannot_lose_the_line():
    """One rank raises in the MIDDLE of a side block's timed loop (after its warm-up), another block raises
    at its start on the other rank: no rank may be left waiting in a barrier (the run ends, rc 0), the
    headline is intact, the two blocks carry an error entry and the blocks after them still ran."""
    d, res = _two_ranks({"MPDATA_BENCH_FAIL": "tracer_batched:1:mid,reference_layout_device_call:0"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["frac"] > 0 and d["steps"] == 3
    assert "error" in d["tracer_batched"] and "value" not in d["tracer_batched"]
    assert "InjectedFailure" in d["reference_layout_device_call"]["error"]
    assert d["twice_the_instances"]["value"] > 0 and d["step_with_fresh_uw"]["value"] > 0
    assert "scatter_gather" in d


@pytest.mark.gpu
def test_headline_only():
    d, _ = _two_ranks(extra_args=("--headline-only",))
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    for k in ("tracer_batched", "twice_the_instances", "reference_layout_device_call", "fp32", "scatter_gather"):
        assert k not in d


@pytest.mark.gpu
def test_a_rank_that_hangs_in_a_side_block_cannot_lose_the_line():
    """Rank 1 never comes back from a side block (what a hung collective or a stuck kernel looks like from outside):
    rank 0 finishes the block and waits in the ranks' agreement.  The watchdog (bench.py class Lifeline) ends the run
    at the block's deadline: the line is printed with the headline and every block measured before, an error entry
    under the block's name, exit code 0 on every rank."""
    d, res = _two_ranks({"MPDATA_BENCH_FAIL": "reference_layout_device_call:1:hang"}, ("--block-timeout", "25"))
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert d["tracer_batched"]["value"] > 0 and d["step_with_fresh_uw"]["value"] > 0
    assert "watchdog" in d["reference_layout_device_call"]["error"]
    assert "BENCH_WATCHDOG" in res.stderr


@pytest.mark.gpu
def test_a_rank_that_dies_cannot_lose_the_line():
    """Rank 1 is killed outright inside a side block (SIGKILL, as the out-of-memory killer or a GPU fault would):
    torch.distributed.run then sends SIGTERM to rank 0, which may sit in a collective at that moment.  The run fails
    (exit code != 0) but rank 0 still prints the line with everything measured until then."""
    d, res = _two_ranks({"MPDATA_BENCH_FAIL": "tracer_batched:1:kill"}, rc0=False)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["frac"] > 0 and d["steps"] == 3
    assert d["step_with_fresh_uw"]["value"] > 0 and d["twice_the_instances"]["value"] > 0
    # (gloo notices the lost peer by itself and the agreement raises; RCCL would wait until the SIGTERM arrives)
    assert "SIGTERM" in d.get("terminated", "") or "agreement" in d["tracer_batched"]["error"]
    assert "value" not in d.get("tracer_batched", {})
