#!/bin/bash
# Socket power and shader clock (rocm-smi, 4 samples / s) while the side kernels loop for seconds:
# biharmonic_wk_scalar on live data and on a decayed field, the flux nest with local / random connectivity.
# -> gpurun_out/power_side.txt (summary lines: tools/power_summary.py)
OUT=${1:-gpurun_out/power_side.txt}
sample() {
  while [ ! -e /tmp/ps_stop ]; do
    echo "== $1 $(date +%s.%N)"
    rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "power \(W\)|sclk" | head -4
    sleep 0.25
  done
}
run() {
  rm -f /tmp/ps_stop
  sample "$1" >> $OUT &
  SP=$!
  shift
  "$@" 2>&1 | grep -v amdgpu.ids | tail -4 >> $OUT
  touch /tmp/ps_stop; wait $SP
}
: > $OUT
run bwk_live_fast python3 tools/bwk_live.py --seconds 5
run bwk_live_exact python3 tools/bwk_live.py --seconds 5 --variant exact
run bwk_decayed_fast python3 tools/bwk_live.py --seconds 5 --decayed
run nlk_local_and_random python3 tools/nlk_bench.py 3000
rm -f /tmp/ps_stop
