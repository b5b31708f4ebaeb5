#!/usr/bin/env python3
"""Rewrites the record table of DESIGN.md section 7 from profiles/<tag>_bench_driver.json (the ONE line bench.py printed
with the driver's arguments on the final library of the round).  usage: python tools/design_table.py [r05]"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
d = json.loads(open(os.path.join(root, "profiles", f"{tag}_bench_driver.json")).read().strip().splitlines()[-1])


def row(b):
    r = b["roofline"]
    return "%.4f" % b["ms_per_step"], "%.3f" % r["frac"], "%.1f" % (b["value"] / 1e9)


ex, tf = d["exact_variant"], d["two_launches_in_flight"]
t1 = row(d)
tbl = ("| workload | ms / step | frac of 8 TB/s | Gcu/s |\n|---|---|---|---|\n"
       "| configs[2] one tracer, FAST, cold (headline) | %s | **%s** | %s |\n" % t1 +
       "| configs[3] 25 tracers, FAST | %s | %s | %s |\n" % row(d["tracer_batched"]) +
       "| `mpdata_plan_run_uw` | %s | %s | %s |\n" % row(d["step_with_fresh_uw"]) +
       "| reference-layout device call | %s | %s | %s |\n" % row(d["reference_layout_device_call"]) +
       "| fp32 plan | %s | %s | %s |\n" % row(d["fp32"]) +
       ("| 72 levels (%s) | %s | %s | %s |\n" % ((d["levels_above_64"]["workload"].split(" at ")[1].split(" fp64")[0],) + row(d["levels_above_64"]))
        if "levels_above_64" in d else "") +
       "| EXACT: one tracer / 25 tracers / device call | %.3f / %.2f / %.3f | %.3f / %.3f / %.3f | %.1f / %.1f / %.1f |\n" % (
           ex["ms_per_step"], ex["tracer_batched"]["ms_per_step"], ex["reference_layout_device_call"]["ms_per_step"],
           ex["roofline"]["frac"], ex["tracer_batched"]["roofline"]["frac"], ex["reference_layout_device_call"]["roofline"]["frac"],
           ex["value"] / 1e9, ex["tracer_batched"]["value"] / 1e9, ex["reference_layout_device_call"]["value"] / 1e9) +
       "| two launches in flight (two streams; throughput only, NOT the headline) | %.4f | %.3f | %.1f |\n" % (
           tf["ms_per_step"], tf["frac_of_8TBs_throughput"], tf["value"] / 1e9) +
       "| reference executable, 1 host core | — | — | %.3f |" % (d["cpu_baseline"]["value"] / 1e9))
path = os.path.join(root, "DESIGN.md")
s = open(path).read()
i = s.index("| workload | ms / step | frac of 8 TB/s | Gcu/s |")
j = s.index("\n\n", i)
s = s[:i] + tbl + s[j:]
i = s.index("Round-5 record (builder's box")
j = s.index("\n\n", i)
r = d["roofline"]
s = s[:i] + ("Round-5 record (builder's box, the FINAL library, driver arguments `--gpus 1 --steps 20 --warmup 5`;\n"
             "profiles/%s_bench_driver.json): the timed average sits %.1f %% %s the per-launch median, `frac_of_measured_ceiling` %.3f;\n"
             "kernel trace and PMC passes: %s_kernel_stats.csv, %s_pmc_summary.json.  Boxes differ by 2–3 %% on one binary.  Other\n"
             "shapes (%s_shape_sweep.txt; nx 32 … 256, nz 28 … 110): FAST 0.62–0.70, nz = 125 0.54." % (
                 tag, abs(r["avg_over_median"] - 1) * 100, "above" if r["avg_over_median"] > 1 else "below",
                 r["frac_of_measured_ceiling"], tag, tag, tag)) + s[j:]
open(path, "w").write(s)
print(tbl)
