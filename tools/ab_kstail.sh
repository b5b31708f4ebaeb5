#!/bin/bash
# nz > 64 with a short last window: the tail form (a share of a wave per instance) against a wave of its own
# (MPDATA_KS_TAIL=0), interleaved.  usage (GPU): bash tools/ab_kstail.sh > gpurun_out/ab_kstail.txt
for s in "24576 32 72" "24576 32 66" "20480 32 80" "16384 32 90" "12288 64 72"; do
  set -- $s
  for v in fast exact; do
    for pass in 1 2; do
      for t in 1 0; do
        echo "== ncrms $1 nx $2 nz $3 $v tail=$t pass $pass"
        MPDATA_KS_TAIL=$t python3 tools/uw_bench.py --variant $v --no-uw --no-conv --ncrms $1 --nx $2 --nz $3 --steps 40 --sets 8 2>&1 | grep -E "plan|rror"
      done
    done
  done
done
