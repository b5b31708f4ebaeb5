"""Aggregate gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into profiles/:
<tag>_kernel_stats.csv, <tag>_domain_stats.csv, <tag>_pmc_summary.json, hbm_traffic.json."""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
KERNEL = "mpdata_advect_xmarch_kernel"
NCRMS, NX, NZ = 65536, 32, 28


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if len(hits) != 1:
        raise SystemExit(f"expected one file for {pattern}, found {hits}")
    return hits[0]


def counters(passname):
    """{counter: (sum over dispatches of the kernel, number of dispatches)}"""
    acc, disp = {}, {}
    with open(one(f"{passname}/**/*_counter_collection.csv")) as fh:
        for row in csv.DictReader(fh):
            if KERNEL not in row["Kernel_Name"]:
                continue
            c = row["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
            disp.setdefault(c, set()).add(row["Dispatch_Id"])
    return {c: (v, len(disp[c])) for c, v in acc.items()}


shutil.copy(one("kt/**/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(one("kt/**/*_domain_stats.csv"), os.path.join(dst, f"{tag}_domain_stats.csv"))
avg_ns = None
with open(os.path.join(dst, f"{tag}_kernel_stats.csv")) as fh:
    for row in csv.DictReader(fh):
        if KERNEL in row["Name"]:
            avg_ns = float(row["AverageNs"]); calls = int(row["Calls"]); name = row["Name"]

fetch, write, sq = counters("fetch"), counters("write"), counters("sq")
nd = fetch["FETCH_SIZE"][1]
# gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
# FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports half of wide coalesced reads.
rd = fetch["FETCH_SIZE"][0] / nd * 1024 * 2
wr = write["WRITE_SIZE"][0] / write["WRITE_SIZE"][1] * 1024
alg = 8 * (NZ - 1) * (4 * NX + 23) * NCRMS
nsteps = NX + 6
waves = sq["SQ_WAVES"][0] / sq["SQ_WAVES"][1]
per = {c: v / n / (waves * nsteps) for c, (v, n) in sq.items() if c != "SQ_WAVES"}
summary = {
    "kernel": name, "config": f"ncrms={NCRMS} nx={NX} nz={NZ}, 1 tracer, FAST variant",
    "commands": "tools/profile_round.sh (rocprofv3 --kernel-trace --stats; separate --pmc passes: "
                "FETCH_SIZE | WRITE_SIZE GRBM_GUI_ACTIVE | SQ_*)",
    "kernel_trace_avg_ns": avg_ns, "kernel_trace_calls": calls,
    "correction": "FETCH_SIZE KiB x2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE KiB "
                  "exact; MI355X_MICROARCH.md section HBM",
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
    "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg,
    "ratio_traffic_over_algorithmic": (rd + wr) / alg,
    "sq_per_wave_step_quad_cycles": per,
    "valu_busy_fraction": per["SQ_ACTIVE_INST_VALU"] * 4 / per["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in per else None,
    "note": f"SQ_* per wave and column step ({int(waves)} waves x {nsteps} steps); SQ_ACTIVE_*/WAVE_CYCLES in "
            "quad-cycles; 4 waves share a SIMD, so VALU busy per SIMD = 4 x ACTIVE_INST_VALU / WAVE_CYCLES",
}
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
with open(os.path.join(dst, "hbm_traffic.json"), "w") as fh:
    key = f"ncrms{NCRMS}_nx{NX}_nz{NZ}_t1"
    json.dump({"fast_" + key: {"hbm_bytes_per_launch": rd + wr, "source": f"profiles/{tag}_pmc_summary.json"},
               "exact_" + key: {"hbm_bytes_per_launch": rd + wr,
                                "source": f"profiles/{tag}_pmc_summary.json (same loads/stores in both variants)"}},
              fh, indent=1)
print(json.dumps(summary, indent=1))

# ---- second kernel: biharmonic_wk_scalar (tools/bwk_bench.py: nelemd=5400, both variants) ----
if glob.glob(os.path.join(src, "bwk_kt", "**", "*_kernel_stats.csv"), recursive=True):
    KERNEL = "bwk_kernel"
    shutil.copy(one("bwk_kt/**/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_bwk_kernel_stats.csv"))
    rows = {}
    with open(os.path.join(dst, f"{tag}_bwk_kernel_stats.csv")) as fh:
        for row in csv.DictReader(fh):
            if KERNEL in row["Name"]:
                rows[row["Name"]] = {"avg_ns": float(row["AverageNs"]), "calls": int(row["Calls"])}
    bf, bw = counters("bwk_fetch"), counters("bwk_write")
    brd = bf["FETCH_SIZE"][0] / bf["FETCH_SIZE"][1] * 1024 * 2
    bwr = bw["WRITE_SIZE"][0] / bw["WRITE_SIZE"][1] * 1024
    balg = 2 * 8 * 16 * 72 * 40 * 5400 + 8 * (144 * 5400 + 16)
    bsum = {"kernel": "bwk_kernel (biharmonic_wk_scalar), nelemd=5400 nlev=72 qsize=40", "kernel_trace": rows,
            "hbm_read_bytes_per_launch": brd, "hbm_write_bytes_per_launch": bwr,
            "algorithmic_bytes_per_launch": balg, "ratio_traffic_over_algorithmic": (brd + bwr) / balg,
            "commands": "tools/profile_round.sh (python3 tools/bwk_bench.py --child -)"}
    with open(os.path.join(dst, f"{tag}_bwk_summary.json"), "w") as fh:
        json.dump(bsum, fh, indent=1)
    print(json.dumps(bsum, indent=1))
