#!/bin/bash
# The records of a round, on the GPU box:  bash tools/record_round.sh r04
#   gpurun_out/<tag>_final/bench_driver.json   bench.py with the driver's arguments (--steps 20 --warmup 5)
#   gpurun_out/<tag>_final/bench_default.json  bench.py with its defaults (100 + 100 launches)
#   gpurun_out/<tag>_final/fortran_driver.txt  the Fortran driver: host mode (1 tracer), device mode (25 tracers),
#                                              ngpus = 1 through the multi-GPU path, 2 shards on one device
#   gpurun_out/<tag>_final/fortran_side_drivers.txt  bwk_driver and nested_hip (second / third kernel)
#   gpurun_out/<tag>_final/shape_sweep.txt, ab_kstail.txt   tools/shape_sweep.sh, tools/ab_kstail.sh
#   gpurun_out/prof_<tag>/                      tools/profile_round.sh (rocprofv3 kernel trace + PMC passes)
# then here: python tools/pmc_summary.py <tag>; copy the records into profiles/.
TAG=${1:-r05}
OUT=gpurun_out/${TAG}_final
mkdir -p $OUT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver rc=$?"
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
F=codesign-kernels_amd/fortran/advect
{
  echo "== host mode, 1 tracer, FAST:  $F 65536 32 28 1 1 - - 1 1 host"
  $F 65536 32 28 1 1 - - 1 1 host
  echo "== device mode, 1 tracer, FAST:  $F 65536 32 28 1 1 - - 1 1 device"
  $F 65536 32 28 1 1 - - 1 1 device
  echo "== device mode, 25 tracers (BASELINE configs[3]), FAST:  $F 65536 32 28 1 1 - - 25 1 device"
  $F 65536 32 28 1 1 - - 25 1 device
  echo "== ngpus = 2 on ONE device (MPDATA_MULTI_DEVICES=0,0: peer-copy transport), device mode, 25 tracers, ncrms = 131072"
  MPDATA_MULTI_DEVICES=0,0 $F 131072 32 28 1 1 - - 25 2 device
  echo "== ngpus = 2 on ONE device, host mode (direct transport), 2 tracers, ncrms = 65536"
  MPDATA_MULTI_DEVICES=0,0 $F 65536 32 28 1 1 - - 2 2 host
  echo "== 72 levels (several waves per instance; the device call goes through the calling thread's plan), device mode, FAST:  $F 24576 32 72 1 1 - - 1 1 device"
  $F 24576 32 72 1 1 - - 1 1 device
  echo "== namelist: configs[3] through a namelist file"
  printf "&advect_nml ncrms=65536, nx=32, nz=28, dist=1, variant=1, ntracers=25, ngpus=1, mode='device' /\n" > $OUT/case.nml
  $F $OUT/case.nml
} > $OUT/fortran_driver.txt 2>&1
echo "fortran rc=$?"
# the drivers of the second and third kernel (the reference's programs of atmosphere/ and nested_loops/)
D=codesign-kernels_amd/fortran
{
  echo "== $D/bwk_driver 16 0   (the reference's shipped size, EXACT)"; $D/bwk_driver 16 0
  echo "== $D/bwk_driver 5400 1 (a cubed-sphere ne=30 mesh, FAST; host arrays: transfers included)"; $D/bwk_driver 5400 1
  echo "== $D/nested_hip - 1    (the reference's shipped namelist, FAST)"; $D/nested_hip - 1
  printf "&nested_nml\n nIters = 20\n nEdges = 819200\n nCells = 89600\n nVertLevels = 100\n nAdv = 10\n/\n" > $OUT/nested32.nml
  echo "== $D/nested_hip nested32.nml 1   (a mesh 32 x the namelist's, random connectivity, FAST)"; $D/nested_hip $OUT/nested32.nml 1
} > $OUT/fortran_side_drivers.txt 2>&1
echo "side drivers rc=$?"
# the plan kernel over the CRM shapes in use (cold, one tracer), and the tail form of nz 65 .. 90 against whole-wave windows
bash tools/shape_sweep.sh fast > $OUT/shape_sweep.txt 2>&1; bash tools/shape_sweep.sh exact | sed 's/^== /== EXACT /' >> $OUT/shape_sweep.txt 2>&1; echo "shape sweep rc=$?"
bash tools/ab_kstail.sh > $OUT/ab_kstail.txt 2>&1; echo "kstail rc=$?"
# the profile passes take ~10 minutes: a gpurun call of their own (PROFILE=0 skips them here)
if [ "${PROFILE:-1}" = 1 ]; then timeout -k 10 1000 bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1; echo "profile rc=$?"; fi
