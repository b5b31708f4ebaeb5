"""Aggregate gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into profiles/:
<tag>_kernel_stats.csv, <tag>_domain_stats.csv, <tag>_t25_kernel_stats.csv, <tag>_pmc_summary.json,
hbm_traffic.json (the per-launch HBM bytes bench.py quotes as `roofline.traffic`, with their source)."""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
NCRMS, NX, NZ = 65536, 32, 28
# kernel-name fragments: wave-major kernel, column-granular fetch (1 tracer) / pair-granular
# (tracer batches); x-march kernel (reference layout)
# (template arguments: R, LPS, WPB, STREAM, tracers per wave)
K_WM1, K_WMT, K_XM = "mpdata_advect_wm_kernel<double,32,4,true", "mpdata_advect_wm_kernel<double,32,4,false", "mpdata_advect_xmarch_kernel"
K_WMX = "mpdata_advect_wm_kernel<double,32,8,true,1,true"   # u, w from the reference layout (mpdata_plan_run_uw)
K_WMO = "mpdata_advect_wm_odd_kernel<double,32,4"           # round 5: tracer batches with an odd count, ONE launch


def is_k(kernel, name):
    """kernel-name match on the demangled name, blanks ignored (mangled fragments also accepted)"""
    n = name.replace(" ", "")
    if "mpdata_exact" in n or "12mpdata_exact" in n:   # (bench.py's exact_variant block launches the EXACT instantiations of the
        return False                                   #  same templates: the records are the FAST kernels')
    alt = {K_WM1: "wm_kernelIdLi32ELi4ELb1", K_WMT: "wm_kernelIdLi32ELi4ELb0", K_WMX: "wm_kernelIdLi32ELi8ELb1ELi1ELb1",
           K_WMO: "wm_odd_kernelIdLi32ELi4E"}.get(kernel, kernel)
    return kernel in n or alt in n


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if len(hits) != 1:
        raise SystemExit(f"expected one file for {pattern}, found {hits}")
    return hits[0]


def have(pattern):
    return len(glob.glob(os.path.join(src, pattern), recursive=True)) == 1


def full_size_rows(passname, kernel):
    """rows of the dispatches of `kernel` with the LARGEST grid of the pass: a side block that launches the
    same kernel on a smaller problem (round 2: the chunked host call) must not enter a per-launch mean"""
    rows = []
    with open(one(f"{passname}/**/*_counter_collection.csv")) as fh:
        rows = [r for r in csv.DictReader(fh) if is_k(kernel, r["Kernel_Name"])]
    if not rows:
        raise SystemExit(f"no dispatch of {kernel} in pass {passname}")
    gmax = max(int(r["Grid_Size"]) for r in rows)
    return [r for r in rows if int(r["Grid_Size"]) == gmax]


def counters(passname, kernel):
    """{counter: mean over the full-size dispatches of `kernel`} (a counter's rows of one dispatch are summed)"""
    acc, disp = {}, {}
    for row in full_size_rows(passname, kernel):
        c = row["Counter_Name"]
        acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
        disp.setdefault(c, set()).add(row["Dispatch_Id"])
    return {c: v / len(disp[c]) for c, v in acc.items()}


def durations(passname, kernel):
    """mean End-Start [ns] of the dispatches of `kernel` in a counter pass"""
    seen = {}
    for row in full_size_rows(passname, kernel):
        if "End_Timestamp" in row:
            seen[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return sum(seen.values()) / len(seen) if seen else None


def trace(passname, kernel, out):
    shutil.copy(one(f"{passname}/**/*_kernel_stats.csv"), os.path.join(dst, out))
    with open(os.path.join(dst, out)) as fh:
        for row in csv.DictReader(fh):
            if is_k(kernel, row["Name"]):
                r = {"name": row["Name"], "avg_ns": float(row["AverageNs"]), "calls": int(row["Calls"]),
                     "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
                r.update(per_dispatch(passname, kernel))
                return r
    return None


def per_dispatch(passname, kernel):
    """median and the mean of the LAST 40 dispatches of `kernel` in a kernel-trace pass (the bench process starts
    from an idle chip: its first launches run 10 % faster than the steady state the timed region sees, and the
    average over all dispatches contains them)"""
    if not have(f"{passname}/**/*_kernel_trace.csv"):
        return {}
    d = []
    with open(one(f"{passname}/**/*_kernel_trace.csv")) as fh:
        for row in csv.DictReader(fh):
            if is_k(kernel, row["Kernel_Name"]):
                d.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    d = [x[1] for x in sorted(d)]
    if len(d) < 8:
        return {}
    last = d[-40:]
    return {"median_ns": float(sorted(d)[len(d) // 2]), "mean_last40_ns": sum(last) / len(last),
            "mean_first5_ns": sum(d[:5]) / 5.0}


def alg_bytes(t):
    return NCRMS * 8 * (NZ - 1) * (t * (2 * NX + 11) + 2 * NX + 12)


def hbm(fetch_pass, write_pass, kernel):
    # gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
    # FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports half of wide coalesced reads.
    rd = counters(fetch_pass, kernel)["FETCH_SIZE"] * 1024 * 2
    wr = counters(write_pass, kernel)["WRITE_SIZE"] * 1024
    return rd, wr


def check_traffic(name, traffic_bytes, algorithmic):
    """HBM traffic below the algorithmic bytes is impossible: it means the wrong dispatches were averaged"""
    if traffic_bytes < 0.98 * algorithmic:
        raise SystemExit(f"{name}: measured traffic {traffic_bytes:.4g} B < algorithmic {algorithmic:.4g} B -- "
                         "the per-launch mean mixes dispatches of different sizes")


summary = {"commands": "tools/profile_round.sh: rocprofv3 --kernel-trace --stats and separate --pmc passes "
                       "(FETCH_SIZE | WRITE_SIZE | TCC_* | SQ_*) of bench.py",
           "correction": "FETCH_SIZE KiB x 2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE KiB exact; "
                         "MI355X_MICROARCH.md section HBM"}
traffic = {}

# ---- headline: 1 tracer, plan API, wave-major layout ------------------------------------------
kt = trace("kt", K_WM1, f"{tag}_kernel_stats.csv")
shutil.copy(one("kt/**/*_domain_stats.csv"), os.path.join(dst, f"{tag}_domain_stats.csv"))
rd, wr = hbm("fetch", "write", K_WM1)
check_traffic("t1_wavemajor", rd + wr, alg_bytes(1))
sq = counters("sq", K_WM1)
waves = sq["SQ_WAVES"]
summary["t1_wavemajor"] = {
    "config": f"ncrms={NCRMS} nx={NX} nz={NZ}, 1 tracer, FAST variant, plan API (wave-major layout)",
    "kernel_trace": kt, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
    "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg_bytes(1),
    "ratio_traffic_over_algorithmic": (rd + wr) / alg_bytes(1),
    "algorithmic_GBs_at_kernel_trace_avg": alg_bytes(1) / kt["avg_ns"],
    # since round 4 EVERY dispatch of the profiled bench run is cold (the wake-up launches cycle through the scratch field
    # sets too), so the plain average of the kernel trace no longer flatters the kernel -- it now UNDERSTATES the timed
    # region: it contains the ~150 wake-up launches that absorb the clock / power-state ramp after idle (0.40-0.43 ms
    # each).  The last 40 dispatches are the warm-up, timed and per-launch passes of bench.py.
    "frac_of_8TBs": alg_bytes(1) / kt["avg_ns"] / 8000.0,
    "frac_of_8TBs_last40_dispatches": alg_bytes(1) / kt.get("mean_last40_ns", kt["avg_ns"]) / 8000.0,
    "sq_per_wave": {c: v / waves for c, v in sq.items() if c != "SQ_WAVES"}, "waves": waves,
    "valu_instructions_per_wave": sq["SQ_INSTS_VALU"] / waves}
traffic[f"fast_ncrms{NCRMS}_nx{NX}_nz{NZ}_t1_wm"] = {"hbm_bytes_per_launch": rd + wr,
                                                     "source": f"profiles/{tag}_pmc_summary.json t1_wavemajor"}
traffic[f"exact_ncrms{NCRMS}_nx{NX}_nz{NZ}_t1_wm"] = {"hbm_bytes_per_launch": rd + wr,
                                                      "source": "same loads / stores in both variants"}

# ---- 25 tracers: one plan run = the batch kernel on 24 tracers (two per wave) + the one-tracer
#      kernel on the last one (same kernel and size as the headline launches) -------------------
def has_kernel(passname, kernel):
    with open(one(f"{passname}/**/*_kernel_stats.csv")) as fh:
        return any(is_k(kernel, row["Name"]) for row in csv.DictReader(fh))


if have("t25_fetch/**/*_counter_collection.csv"):
    ONE_LAUNCH = has_kernel("t25_kt", K_WMO)   # round 5: the odd tracer rides in the batch launch
    if ONE_LAUNCH:
        K_WMT = K_WMO
        kt25 = trace("t25_kt", K_WMT, f"{tag}_t25_kernel_stats.csv")
        kt25_1 = None
        run_ns = kt25["avg_ns"]
        rd, wr = hbm("t25_fetch", "t25_write", K_WMT)
    else:
        kt25 = trace("t25_kt", K_WMT, f"{tag}_t25_kernel_stats.csv")
        kt25_1 = trace("t25_kt", K_WM1, f"{tag}_t25_kernel_stats.csv")
        run_ns = kt25["avg_ns"] + kt25_1["avg_ns"]
        rd, wr = hbm("t25_fetch", "t25_write", K_WMT)
        rd1, wr1 = hbm("t25_fetch", "t25_write", K_WM1)
        rd, wr = rd + rd1, wr + wr1
    check_traffic("t25_wavemajor", rd + wr, alg_bytes(25))
    tcc = counters("t25_tcc", K_WMT)
    sq = counters("t25_sq", K_WMT)
    waves = sq["SQ_WAVES"]
    vpw = sq["SQ_INSTS_VALU"] / waves
    valu = sq["SQ_INSTS_VALU"] + (0.0 if ONE_LAUNCH else counters("t25_sq", K_WM1)["SQ_INSTS_VALU"])
    valu_peak = 33.0e12 / 64.0   # wave-instructions / s, tools/valu_rate.hip
    summary["t25_wavemajor"] = {
        "config": f"ncrms={NCRMS} nx={NX} nz={NZ}, 25 tracers, FAST variant, plan API (wave-major layout)",
        "kernels_per_run": "ONE launch: 12 two-tracer waves + 1 one-tracer wave per tile (mpdata_advect_wm_odd_kernel)" if ONE_LAUNCH
                           else "batch kernel (24 tracers, two per wave) + one-tracer kernel (tracer 25)",
        "kernel_trace": kt25, "kernel_trace_last_tracer": kt25_1, "run_ns": run_ns,
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
        "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg_bytes(25),
        "ratio_traffic_over_algorithmic": (rd + wr) / alg_bytes(25),
        "algorithmic_GBs_at_kernel_trace_avg": alg_bytes(25) / run_ns,
        "hbm_frac_of_8TBs": alg_bytes(25) / run_ns / 8000.0,
        "l2_hit_rate": tcc["TCC_HIT_sum"] / (tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"]),
        "waves": waves, "valu_instructions_per_wave": vpw,
        "valu_instructions_per_launch": valu,
        "valu_time_at_measured_peak_ms": valu / valu_peak * 1e3,
        "valu_frac": valu / valu_peak / (run_ns * 1e-9),
        "valu_note": "VALU wave-instructions per plan run (both kernels) / (33e12 lane-ops/s / 64, tools/valu_rate.hip) "
                     "/ run time; the peak was measured at the 2.0 GHz the microbenchmark sustains, the batch kernel "
                     "runs at the clock below (power-limited)",
        "sq_per_wave": {c: v / waves for c, v in sq.items() if c != "SQ_WAVES"}}
    gui = counters("t25_write", K_WMT).get("GRBM_GUI_ACTIVE")
    if gui:   # cycles summed over the 8 XCDs / kernel time of the same (serialised) dispatch
        t_ns = durations("t25_write", K_WMT)
        if t_ns:
            summary["t25_wavemajor"]["clock_GHz"] = gui / 8.0 / t_ns
    for v in ("fast", "exact"):
        traffic[f"{v}_ncrms{NCRMS}_nx{NX}_nz{NZ}_t25_wm"] = {"hbm_bytes_per_launch": rd + wr,
                                                            "source": f"profiles/{tag}_pmc_summary.json t25_wavemajor"}

# ---- reference-layout device call (x-march kernel) --------------------------------------------
if have("ref_fetch/**/*_counter_collection.csv"):
    rd, wr = hbm("ref_fetch", "ref_write", K_XM)
    check_traffic("t1_reference_layout", rd + wr, alg_bytes(1))
    summary["t1_reference_layout"] = {
        "config": f"ncrms={NCRMS} nx={NX} nz={NZ}, 1 tracer, FAST variant, mpdata_advect_scalar2d_device (x-march kernel)",
        "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg_bytes(1),
        "ratio_traffic_over_algorithmic": (rd + wr) / alg_bytes(1)}
    for v in ("fast", "exact"):
        traffic[f"{v}_ncrms{NCRMS}_nx{NX}_nz{NZ}_t1"] = {"hbm_bytes_per_launch": rd + wr,
                                                         "source": f"profiles/{tag}_pmc_summary.json t1_reference_layout"}

# ---- mpdata_plan_run_uw: f in the plan layout, u and w fetched from the reference layout ----------
if have("uw_fetch/**/*_counter_collection.csv"):
    ktu = trace("uw_kt", K_WMX, f"{tag}_uw_kernel_stats.csv")
    rd, wr = hbm("uw_fetch", "uw_write", K_WMX)
    check_traffic("t1_run_uw", rd + wr, alg_bytes(1))
    summary["t1_run_uw"] = {
        "config": f"ncrms={NCRMS} nx={NX} nz={NZ}, 1 tracer, FAST variant, mpdata_plan_run_uw (u, w reference layout, f plan layout)",
        "kernel_trace": ktu, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
        "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg_bytes(1),
        "ratio_traffic_over_algorithmic": (rd + wr) / alg_bytes(1),
        "frac_of_8TBs": alg_bytes(1) / ktu["avg_ns"] / 8000.0 if ktu else None}
    traffic[f"fast_ncrms{NCRMS}_nx{NX}_nz{NZ}_t1_uw"] = {"hbm_bytes_per_launch": rd + wr,
                                                        "source": f"profiles/{tag}_pmc_summary.json t1_run_uw"}

with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
with open(os.path.join(dst, "hbm_traffic.json"), "w") as fh:
    json.dump(traffic, fh, indent=1)
print(json.dumps(summary, indent=1))

# ---- second kernel: biharmonic_wk_scalar (tools/bwk_bench.py: nelemd=5400, both variants) ----
if have("bwk_kt/**/*_kernel_stats.csv"):
    KERNEL = "bwk_kernel"
    shutil.copy(one("bwk_kt/**/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_bwk_kernel_stats.csv"))
    rows = {}
    with open(os.path.join(dst, f"{tag}_bwk_kernel_stats.csv")) as fh:
        for row in csv.DictReader(fh):
            if is_k(KERNEL, row["Name"]):
                rows[row["Name"]] = {"avg_ns": float(row["AverageNs"]), "calls": int(row["Calls"])}
    brd = counters("bwk_fetch", KERNEL)["FETCH_SIZE"] * 1024 * 2
    bwr = counters("bwk_write", KERNEL)["WRITE_SIZE"] * 1024
    balg = 2 * 8 * 16 * 72 * 40 * 5400 + 8 * (144 * 5400 + 16)
    bsum = {"kernel": "bwk_kernel (biharmonic_wk_scalar), nelemd=5400 nlev=72 qsize=40", "kernel_trace": rows,
            "hbm_read_bytes_per_launch": brd, "hbm_write_bytes_per_launch": bwr,
            "algorithmic_bytes_per_launch": balg, "ratio_traffic_over_algorithmic": (brd + bwr) / balg,
            "commands": "tools/profile_round.sh (python3 tools/bwk_bench.py --child -)"}
    with open(os.path.join(dst, f"{tag}_bwk_summary.json"), "w") as fh:
        json.dump(bsum, fh, indent=1)
    print(json.dumps(bsum, indent=1))

# ---- third kernel: high-order flux nest on the 32 x mesh (tools/nlk_bench.py: local, then random connectivity)
if have("nlk_kt/**/*_kernel_stats.csv"):
    KERNEL = "nlk_kernel"
    shutil.copy(one("nlk_kt/**/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_nlk_kernel_stats.csv"))
    per = {}
    for which in ("nlk_fetch", "nlk_write"):
        for row in full_size_rows(which, KERNEL):
            per.setdefault(row["Dispatch_Id"], {}).setdefault(row["Counter_Name"], 0.0)
            per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
    # dispatch order = tools/nlk_bench.py's order: first half local connectivity, second half random
    def mean(ids, name, scale):
        v = [per[i][name] for i in ids if name in per[i]]
        return sum(v) / len(v) * scale if v else None
    ids = sorted(per, key=int)
    fe = [i for i in ids if "FETCH_SIZE" in per[i]]; wr_ = [i for i in ids if "WRITE_SIZE" in per[i]]
    half_f, half_w = len(fe) // 2, len(wr_) // 2
    nsum = {"kernel": "nlk_kernel, nEdges=819200 nCells=89600 nVertLevels=100 nAdv=10",
            "commands": "tools/profile_round.sh (python3 tools/nlk_bench.py: N launches local connectivity, then N random)",
            "local_connectivity": {"hbm_read_bytes_per_launch": mean(fe[:half_f], "FETCH_SIZE", 2048),
                                   "hbm_write_bytes_per_launch": mean(wr_[:half_w], "WRITE_SIZE", 1024)},
            "random_connectivity": {"hbm_read_bytes_per_launch": mean(fe[half_f:], "FETCH_SIZE", 2048),
                                    "hbm_write_bytes_per_launch": mean(wr_[half_w:], "WRITE_SIZE", 1024)}}
    with open(os.path.join(dst, f"{tag}_nlk_summary.json"), "w") as fh:
        json.dump(nsum, fh, indent=1)
    print(json.dumps(nsum, indent=1))
