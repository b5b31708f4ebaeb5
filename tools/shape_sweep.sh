#!/bin/bash
# The plan kernel over the shapes E3SM-MMF runs its CRMs at (FAST, one tracer, cold: a plan of its own per timed step),
# about the headline's cell count each.  usage (GPU): bash tools/shape_sweep.sh [variant] > gpurun_out/shape_sweep.txt
V=${1:-fast}
for s in "65536 32 28" "32768 64 28" "32768 32 58" "16384 64 58" "8192 128 58" "4096 256 58" "32768 32 50" "24576 32 72" "12288 64 72" "6144 128 72" "16384 32 110" "8192 64 125"; do
  set -- $s
  echo "== ncrms $1 nx $2 nz $3"
  python3 tools/uw_bench.py --variant $V --no-uw --no-conv --ncrms $1 --nx $2 --nz $3 --steps 40 --sets 10 2>&1 | grep -E "plan|layout|Error|error"
done
