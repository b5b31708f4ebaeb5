#!/bin/bash
# mpdata_plan_run_uw A/B (cold): the kernels that read the caller's u, w against MPDATA_RUN_UW=import (one fused
# conversion pass, then the ordinary kernels): 25-tracer batches at configs[3], one tracer at nz = 58
for pass in 1 2; do
  echo "== pass $pass, 25 tracers, ring"; python3 tools/uw_bench.py --no-plan --no-uw --no-conv --sets 6 --batch 25 | grep "T=25"
  echo "== pass $pass, 25 tracers, import"; MPDATA_RUN_UW=import python3 tools/uw_bench.py --no-plan --no-uw --no-conv --sets 6 --batch 25 | grep "T=25"
done
for pass in 1 2; do
  echo "== pass $pass, nz=58 ncrms=32768, ring"; python3 tools/uw_bench.py --nz 58 --ncrms 32768 --no-conv --steps 40 --sets 10 | grep "T=1"
  echo "== pass $pass, nz=58 ncrms=32768, import"; MPDATA_RUN_UW=import python3 tools/uw_bench.py --nz 58 --ncrms 32768 --no-conv --steps 40 --sets 10 | grep "T=1"
done
