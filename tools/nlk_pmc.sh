#!/bin/bash
# SQ / TA counters of the high-order flux nest on the 32 x mesh (where does a wave wait?).  usage (GPU box): bash tools/nlk_pmc.sh
ROOT=$PWD; OUT=$ROOT/gpurun_out/nlk_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N="python3 $ROOT/tools/nlk_bench.py 3"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -o run -- $N > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -o run -- $N > $OUT/b.log 2>&1
# (a third pass with TA_* / TCP_* derived counters aborted inside rocprofv3 on this pool and hung the call: not run)
cd $ROOT
python3 - <<'PY'
import csv, glob, os
out = os.path.join("gpurun_out", "nlk_pmc")
for p in ("a", "b"):
    f = glob.glob(os.path.join(out, p, "**", "*_counter_collection.csv"), recursive=True)
    if not f:
        print(p, "no csv"); continue
    acc = {}
    for r in csv.DictReader(open(f[0])):
        if "nlk_kernel" not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for c, v in acc.items():
        # per dispatch: a counter has one row per dispatch (or per XCD); print mean of the per-dispatch sums
        print(p, c, "rows", len(v), "mean", sum(v) / len(v))
PY
