"""The recorded bench line (profiles/r04_bench.json, written by bench.py on an MI355X) carries what
the measurement contract asks for: the headline keys, the `roofline` object of the dominant kernel
and the `cpu_baseline` object, with consistent arithmetic.  CPU-only: it reads the committed record."""
import json
import os

import bench_contract as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record():
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def test_recorded_line_meets_the_contract():
    d = _record()
    C.check_headline(d, 65536)
    C.check_roofline(d, 65536)
    C.check_cpu_baseline(d)
    C.check_no_block_failed(d)
    assert d["roofline"]["traffic"] is not None


def test_kernel_trace_agrees_with_the_hip_event_time():
    """Two independent clocks for the same kernel: the rocprofv3 kernel trace of the profiled bench run
    (profiles/r04_kernel_stats.csv; its cold timed region = the last 40 dispatches, profiles/r04_pmc_summary.json) and
    the HIP events of the un-profiled runs (profiles/r04_bench*.json) -- they must agree (within 2 %: box to box)."""
    import csv
    with open(os.path.join(ROOT, "profiles", "r04_pmc_summary.json")) as fh:
        rec = json.load(fh)["t1_wavemajor"]["kernel_trace"]
    name = rec["name"]
    assert "mpdata_advect_wm_kernel<double, 32, 4, true, 1" in name and rec["calls"] >= 60
    with open(os.path.join(ROOT, "profiles", "r04_kernel_stats.csv")) as fh:
        rows = [r for r in csv.DictReader(fh) if r["Name"] == name]
    assert len(rows) == 1 and abs(float(rows[0]["AverageNs"]) - rec["avg_ns"]) < 1.0
    for fn in ("r04_bench.json", "r04_bench_driver.json"):
        with open(os.path.join(ROOT, "profiles", fn)) as fh:
            ms = json.loads(fh.read().strip().splitlines()[-1])["roofline"]["kernel_ms_avg"]
        assert abs(rec["mean_last40_ns"] * 1e-6 - ms) / ms < 0.02, (fn, rec["mean_last40_ns"], ms)
    # every dispatch of the trace is cold (the wake-up launches cycle through the field sets): even the plain average,
    # wake-up launches in the clock ramp included, stays within 3 % of the timed region
    assert abs(rec["avg_ns"] - rec["mean_last40_ns"]) / rec["mean_last40_ns"] < 0.03


def test_tracer_batch_block_reports_both_ceilings():
    t = _record()["tracer_batched"]
    assert "configs[3]" in t["workload"] and t["value"] > 0
    r = t["roofline"]
    assert 0 < r["hbm_frac"] < 1 and 0 < r["valu_frac"] < 1.2
    # BT = 8 nzm (T (2 nx + 11) + 2 nx + 12) per instance
    assert r["algorithmic_bytes_per_launch"] == 65536 * 8 * 27 * (25 * (2 * 32 + 11) + 2 * 32 + 12)


def test_layout_import_of_the_record():
    d = _record()
    lc = d["layout_conversion"]
    # round 4: the import by row segments through LDS-DMA (round 3: 0.28-0.30 / 0.51-0.54 ms)
    assert lc["import_ms_f_per_tracer"] <= 0.23 and lc["import_ms_u_and_w"] <= 0.45


def test_side_blocks_of_round_3():
    d = _record()
    # the round-2 protocol is there as what it is, and is not slower than the cold headline
    sh = d["consecutive_tracers_shared_uw"]
    assert "NOT a cold call" in sh["workload"] and sh["roofline"]["frac"] >= d["roofline"]["frac"] - 0.01
    # one step on fresh reference-layout u, w: the velocities' layout entry is inside the timed region
    fu = d["step_with_fresh_uw"]
    assert "mpdata_plan_run_uw" in fu["workload"] and 0.5 < fu["roofline"]["frac"] < d["roofline"]["frac"]
    assert fu["roofline"]["algorithmic_bytes_per_launch"] == d["roofline"]["algorithmic_bytes_per_launch"]
    lc = d["layout_conversion"]
    assert lc["import_ms_u_and_w"] > 0 and lc["import_ms_f_per_tracer"] > 0 and lc["export_ms_f_per_tracer"] > 0
    r = d["tracer_batched"]["roofline"]
    assert "recorded profile" in r["valu_source"] and 0 < r["binding_floor_frac"] < 1.2
