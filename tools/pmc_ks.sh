#!/bin/bash
# PMC passes over the nz > 64 plan kernel (tail form and plain form): HBM bytes, VALU / wait shares.
# usage (GPU): bash tools/pmc_ks.sh [ncrms nx nz]   -> gpurun_out/pmc_ks/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
N=${1:-24576}; X=${2:-32}; Z=${3:-72}
OUT=$R/gpurun_out/pmc_ks; rm -rf $OUT; mkdir -p $OUT
S="python3 $R/tools/uw_bench.py --no-uw --no-conv --ncrms $N --nx $X --nz $Z --steps 10 --sets 4"
for t in 1 0; do
  export MPDATA_KS_TAIL=$t
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t${t}_trace -o run -- $S > $OUT/t${t}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/t${t}_fetch -o run -- $S > $OUT/t${t}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/t${t}_write -o run -- $S > $OUT/t${t}_write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/t${t}_sq -o run -- $S > $OUT/t${t}_sq.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/t${t}_tcc -o run -- $S > $OUT/t${t}_tcc.log 2>&1
done
python3 - $OUT $N $X $Z > $OUT/summary.txt <<'P'
import csv, glob, sys, collections
out, N, X, Z = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
alg = 8 * (Z - 1) * (4 * X + 23) * N
print("algorithmic bytes per launch", alg)
for t in (1, 0):
    print("== tail form" if t else "== plain form")
    for f in glob.glob(f"{out}/t{t}_trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "advect_wm" in r["Name"]:
                print("  trace:", r["Name"][:70], "calls", r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
    acc = collections.defaultdict(list)
    for p in ("fetch", "write", "sq", "tcc"):
        for f in glob.glob(f"{out}/t{t}_{p}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "advect_wm" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        m = sum(v) / len(v)
        extra = ""
        if k == "FETCH_SIZE": extra = f"  -> {m * 2 * 1024 / 1e9:.3f} GB per launch (x 2 KiB... see guide), {m * 2 * 1024 / alg:.3f} x algorithmic reads+writes"
        if k == "WRITE_SIZE": extra = f"  -> {m * 1024 / 1e9:.3f} GB per launch"
        print(f"  {k:24s} n={len(v):3d} mean {m:.4g}{extra}")
P
cat $OUT/summary.txt
