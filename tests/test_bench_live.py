"""bench.py run live on the GPU (one process, a small shard so that it takes seconds) and its ONE line held against
the measurement contract (tests/bench_contract.py) -- the same checks the recorded line of the round passes."""
import json
import os
import subprocess
import sys

import pytest

import bench_contract as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_live_line_meets_the_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--prewarm-ms", "60", "--prewarm-min-ms", "10", "--ncrms-per-gpu", "8192", "--batched-tracers", "3", "--batched-steps", "2",
           "--no-fp32", "--no-bwk", "--no-host-call"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    # the driver keeps a 10-KB tail of the line for its reader: the line stays compact and the blocks of the hot path
    # come last (side kernels and secondary views first)
    assert len(lines[0]) < 16000, len(lines[0])
    keys = list(d)
    assert keys.index("tracer_batched") > keys.index("consecutive_tracers_shared_uw") > keys.index("layout_conversion")
    assert keys.index("exact_variant") > keys.index("cpu_baseline") > keys.index("consecutive_tracers_shared_uw")
    assert keys[-2] == "tracer_batched" and keys.index("roofline") < keys.index("layout_conversion")
    assert keys[-1] == "headline_repeated" and d["headline_repeated"]["value"] == d["value"] \
        and d["headline_repeated"]["roofline_frac"] == d["roofline"]["frac"]
    C.check_headline(d, 8192)
    C.check_roofline(d, 8192)
    C.check_cpu_baseline(d)
    C.check_no_block_failed(d)
    assert d["steps"] == 4 and d["warmup"] == 2
    w = d["roofline"]["wake_up"]        # the plateau rule ran: whole groups of 8, not longer than its cap + one group
    assert w["groups"] >= 3 and 10.0 <= w["ms_used"] < 200.0 and len(w["group_ms_per_launch"]) >= 3
    assert d["config"]["launched_by"] == "single process"
    # the headline on stderr is the headline of the line
    early = [ln for ln in res.stderr.splitlines() if ln.startswith("BENCH_HEADLINE {")]
    assert len(early) == 1 and json.loads(early[0].split(" ", 1)[1])["value"] == d["value"]
    for blk in ("step_with_fresh_uw", "twice_the_instances", "tracer_batched", "reference_layout_device_call",
                "exact_variant", "levels_above_64"):
        assert d[blk]["value"] > 0 and 0 < d[blk]["roofline"]["frac"] < 1, blk
    tf = d["two_launches_in_flight"]   # throughput only: no per-kernel duration is claimed when two kernels share the chip
    assert tf["value"] > 0 and "roofline" not in tf and 0 < tf["frac_of_8TBs_throughput"] < 1
    ex = d["exact_variant"]     # the variant a default caller gets has a number of its own: plan run, batch, device call
    assert ex["tracer_batched"]["value"] > 0 and ex["reference_layout_device_call"]["value"] > 0
