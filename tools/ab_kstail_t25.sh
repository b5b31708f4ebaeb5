#!/bin/bash
# usage: bash tools/ab_kstail_t25.sh [nz ncrms]
# 25 tracers at 72 levels (default), bench.py's cold protocol: tail form against whole-wave windows (MPDATA_KS_TAIL=0), FAST; interleaved.
mkdir -p gpurun_out; L=gpurun_out/ab_kstail_t25.log; : > $L
NZ=${1:-72}; NC=${2:-24576}
X="--nz $NZ --ncrms-per-gpu $NC --steps 10 --warmup 3 --no-fp32 --no-bwk --no-exact --no-host-call --no-reflayout --no-x2 --no-shared-block --no-fresh-uw --no-cpu-baseline"
for i in 1 2; do
  for t in 1 0; do
    MPDATA_KS_TAIL=$t timeout -k 10 300 python bench.py $X > gpurun_out/ab_kstail_t25.json 2>/dev/null || exit 1
    python - $t $i >> $L <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_kstail_t25.json").read().strip().splitlines()[-1])
t = d["tracer_batched"]; r = t["roofline"]
print("tail=%s pass %s: 25 tracers %.4f ms/step (kernel avg %.4f)  %.1f Gcu/s  hbm_frac %.3f | one tracer %.4f ms frac %.3f"
      % (sys.argv[1], sys.argv[2], t["ms_per_step"], r["kernel_ms_avg"], t["value"] / 1e9, r["frac"],
         d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
PY
  done
done
cat $L
