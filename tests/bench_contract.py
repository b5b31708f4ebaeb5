"""The measurement contract of the ONE JSON line bench.py prints, as checks on a parsed line: used on a line
produced live on the GPU (tests/test_bench_live.py)."""

HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def check_headline(d, ncrms, nx=32, nz=28, ntracers=1, n_gpus=1):
    for k in HEADLINE_KEYS:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    c = d["config"]
    assert "configs[2]" in c["workload"] and "model" not in c
    assert (c["ncrms_per_gpu"], c["nx"], c["nz"], c["ntracers"]) == (ncrms, nx, nz, ntracers)
    # the headline is a COLD measurement and says so: own u, w per timed step, no serpentine tile order
    assert c["uw_shared_across_steps"] is False and c["serpentine"] is False and c["steps_per_field_set"] >= 1
    assert "cycle through the scratch field sets" in c["prewarm"]
    # whole-job throughput = cells per step / seconds per step
    cells = c["ncrms_global"] * c["nx"] * (c["nz"] - 1) * c["ntracers"]
    assert c["ncrms_global"] == ncrms * n_gpus
    assert abs(d["value"] - cells / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def check_roofline(d, ncrms, nx=32, nz=28):
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # 8 * nzm * (4 nx + 23) bytes per instance
    assert r["algorithmic_bytes_per_launch"] == ncrms * 8 * (nz - 1) * (4 * nx + 23)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    # the per-launch list behind min / median is there, and so is what the wake-up did (plateau rule)
    assert len(r["kernel_ms_samples"]) >= 1 and min(r["kernel_ms_samples"]) > 0
    assert abs(min(r["kernel_ms_samples"]) - r["kernel_ms_min"]) < 1e-4
    w = r["wake_up"]
    assert w["cap_ms"] == d["config"]["prewarm_ms_cap"] and w["ms_used"] == d["config"]["prewarm_ms_used"]
    assert w["plateau_reached"] in (True, False, None) and (w["groups"] >= 3 or w["cap_ms"] < 40 or not w["plateau_reached"])
    # the kernel cannot be faster than the step that contains it
    assert r["kernel_ms_avg"] <= d["ms_per_step"] * 1.001
    # HBM bytes from the PMC passes (a recorded profile, for the shapes that have one): at least the algorithmic
    # bytes, labelled with their source -- or null
    if r.get("traffic") is not None:
        assert r["traffic"] >= r["algorithmic_bytes_per_launch"] and "profiles/hbm_traffic.json" in r["traffic_source"]


def check_cpu_baseline(d):
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0


def check_no_block_failed(d):
    for k, v in d.items():
        if isinstance(v, dict):
            assert "error" not in v, (k, v.get("error"))
