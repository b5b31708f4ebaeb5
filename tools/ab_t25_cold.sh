# 25 tracers, COLD protocol of bench.py (a plan per field set), interleaved: odd tracer inside the batch launch (default)
# against a launch of its own behind it (MPDATA_WM_SPLIT=1).  -> gpurun_out/ab_t25_cold.log
mkdir -p gpurun_out; L=gpurun_out/ab_t25_cold.log; : > $L
X="--steps 20 --warmup 5 --no-fp32 --no-bwk --no-exact --no-host-call --no-reflayout --no-x2 --no-shared-block --no-fresh-uw --no-cpu-baseline"
for i in 1 2 3; do
  for v in default split; do
    if [ $v = split ]; then export MPDATA_WM_SPLIT=1; else unset MPDATA_WM_SPLIT; fi
    timeout -k 10 300 python bench.py $X > gpurun_out/ab_t25_cold.json 2>/dev/null || exit 1
    python - $v $i >> $L <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_t25_cold.json").read().strip().splitlines()[-1])
t = d["tracer_batched"]; r = t["roofline"]
print("odd-tracer=%s pass %s: %.4f ms/step (kernel avg %.4f, median %.4f)  %.1f Gcu/s  hbm_frac %.3f | headline %.4f ms frac %.3f"
      % (sys.argv[1], sys.argv[2], t["ms_per_step"], r["kernel_ms_avg"], r["kernel_ms_median"], t["value"] / 1e9, r["frac"],
         d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
PY
  done
done
unset MPDATA_WM_SPLIT
cat $L
