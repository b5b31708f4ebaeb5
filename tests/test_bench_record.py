"""The recorded bench line (profiles/r04_bench.json, written by bench.py on an MI355X) carries what
the measurement contract asks for: the headline keys, the `roofline` object of the dominant kernel
and the `cpu_baseline` object, with consistent arithmetic.  CPU-only: it reads the committed record."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record():
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def test_headline_keys_and_workload():
    d = _record()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    c = d["config"]
    assert "configs[2]" in c["workload"] and (c["ncrms_per_gpu"], c["nx"], c["nz"], c["ntracers"]) == (65536, 32, 28, 1)
    # the headline is a COLD measurement and says so: own u, w per timed step, no serpentine tile order
    assert c["uw_shared_across_steps"] is False and c["serpentine"] is False and c["steps_per_field_set"] >= 1
    # whole-job throughput = cells per step / seconds per step
    cells = c["ncrms_global"] * c["nx"] * (c["nz"] - 1) * c["ntracers"]
    assert abs(d["value"] - cells / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_roofline_object():
    r = _record()["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # 8 * nzm * (4 nx + 23) bytes per instance
    assert r["algorithmic_bytes_per_launch"] == 65536 * 8 * 27 * (4 * 32 + 23)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # HBM bytes from the PMC passes: at least the algorithmic bytes, and labelled with their source
    assert r["traffic"] >= r["algorithmic_bytes_per_launch"] and "profiles/hbm_traffic.json" in r["traffic_source"]
    # the kernel cannot be faster than the step that contains it
    assert r["kernel_ms_avg"] <= _record()["ms_per_step"] * 1.001


def test_cpu_baseline_object():
    c = _record()["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0


def test_tracer_batch_block_reports_both_ceilings():
    t = _record()["tracer_batched"]
    assert "configs[3]" in t["workload"] and t["value"] > 0
    r = t["roofline"]
    assert 0 < r["hbm_frac"] < 1 and 0 < r["valu_frac"] < 1.2
    # BT = 8 nzm (T (2 nx + 11) + 2 nx + 12) per instance
    assert r["algorithmic_bytes_per_launch"] == 65536 * 8 * 27 * (25 * (2 * 32 + 11) + 2 * 32 + 12)


def test_no_block_of_the_record_failed_and_the_wake_up_is_cold():
    d = _record()
    for k, v in d.items():
        if isinstance(v, dict):
            assert "error" not in v, (k, v.get("error"))
    assert "cycle through the scratch field sets" in d["config"]["prewarm"]
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
