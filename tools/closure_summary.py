#!/usr/bin/env python3
"""gpurun_out/closure_r04/ (tools/closure_r04.sh) -> profiles/r04_ablation.json + the kernel-trace CSVs
it was computed from (profiles/r04_abl_*.csv: one row per dispatch of the headline kernel).
The last 60 dispatches of every run are its timed cold launches (uw_bench.py: 60 warm-up + 60 timed,
each on a field set of its own)."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "closure_r04")
ALG = lambda n: 8 * 27 * 151 * n


def dispatches(d):
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if "mpdata_advect_wm_kernel" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def stats(rows, last=60):
    d = [e - s for s, e, _, _ in rows][-last:]
    d.sort()
    return {"dispatches": len(rows), "used_last": len(d), "mean_ns": sum(d) / len(d), "median_ns": d[len(d) // 2],
            "min_ns": d[0], "max_ns": d[-1]}


out = {"protocol": "tools/uw_bench.py --no-uw --no-conv --steps 60 --sets 12 under rocprofv3 --kernel-trace: every launch on "
                   "a plan of its own (12 sets x 1.6 GB), FAST; the last 60 dispatches of a run = its timed launches",
       "algorithmic_bytes_per_launch_65536": ALG(65536), "runs": {}, "sweep": {}}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
for d in sorted(glob.glob(os.path.join(src, "kt_*_[12]"))):
    name = os.path.basename(d)[3:]
    rows = dispatches(d)
    if not rows:
        continue
    st = stats(rows)
    st["frac_of_8TBs"] = ALG(65536) / st["mean_ns"] / 8000.0
    st["kernel"] = rows[-1][3][:120]
    out["runs"][name] = st
    with open(os.path.join(ROOT, "profiles", f"r04_abl_{name}.csv"), "w") as fh:
        fh.write("start_ns,end_ns,duration_ns,grid\n")
        for s, e, g, _ in rows:
            fh.write(f"{s},{e},{e - s},{g}\n")
xs, ys = [], []
for d in sorted(glob.glob(os.path.join(src, "sweep_*")), key=lambda p: int(p.rsplit("_", 1)[1]) if os.path.isdir(p) else 0):
    if not os.path.isdir(d):
        continue
    n = int(d.rsplit("_", 1)[1])
    rows = dispatches(d)
    if not rows:
        continue
    st = stats(rows)
    st["frac_of_8TBs"] = ALG(n) / st["mean_ns"] / 8000.0
    out["sweep"][str(n)] = st
    xs.append(n); ys.append(st["mean_ns"])
if len(xs) >= 3:   # least squares t = a n + b
    import numpy as np
    A = np.vstack([xs, np.ones(len(xs))]).T
    (sl, ic), *_ = np.linalg.lstsq(A, np.array(ys, dtype=float), rcond=None)
    out["sweep_fit"] = {"ns_per_instance": float(sl), "ns_per_65536_instances": float(sl * 65536), "fixed_ns_per_launch": float(ic),
                        "steady_frac_of_8TBs": ALG(65536) / (sl * 65536) / 8000.0}
wt = os.path.join(src, "wave_timeline.json")
if os.path.exists(wt):
    with open(wt) as fh:
        w = json.load(fh)
    out["wave_timeline_median"] = w["median"]
    with open(os.path.join(ROOT, "profiles", "r04_wave_timeline.json"), "w") as fh:
        json.dump(w, fh, indent=1)
with open(os.path.join(ROOT, "profiles", "r04_ablation.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({k: (v if k != "runs" else {a: (round(b["mean_ns"]), round(b["frac_of_8TBs"], 4)) for a, b in v.items()})
                  for k, v in out.items() if k in ("runs", "sweep_fit", "wave_timeline_median")}, indent=1))
