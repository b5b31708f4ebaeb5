#!/bin/bash
# Round-4 closure records of the configs[2] headline: cold kernel-trace CSVs of the
# full kernel, the FIRSTPASS and NOCOMPUTE ablations (wrong results by design: timing only), an ncrms
# sweep of the full kernel, and the per-wave timeline.  Needs the variant builds:
#   bash tools/build_variants.sh "_stamps -DMPDWM_STAMPS" "_nocomp -DMPDWM_ABL_NOCOMPUTE" "_firstpass -DMPDWM_ABL_FIRSTPASS"
# usage (GPU box): bash tools/closure_r04.sh  ->  gpurun_out/closure_r04/ ; then tools/closure_summary.py
set -e
ROOT=$PWD
OUT=$ROOT/gpurun_out/closure_r04
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $ROOT/tools/uw_bench.py --no-uw --no-conv --steps 60 --sets 12"
# interleaved: full / firstpass / nocompute, twice (box drift shows as the difference between the passes)
for pass in 1 2; do
  for v in full firstpass nocomp; do
    if [ $v = full ]; then unset MPDATA_HIP_LIB; else export MPDATA_HIP_LIB=$ROOT/codesign-kernels_amd/libmpdata_hip_$v.so; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${v}_$pass -o run -- $P > $OUT/kt_${v}_$pass.log 2>&1
    echo "done $v $pass: $(grep 'plan   T=1' $OUT/kt_${v}_$pass.log)"
  done
done
unset MPDATA_HIP_LIB
# ncrms sweep of the full kernel (cold; 12 sets each)
for n in 8192 16384 32768 65536 131072 262144; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sweep_$n -o run -- $P --ncrms $n > $OUT/sweep_$n.log 2>&1
  echo "sweep $n: $(grep 'plan   T=1' $OUT/sweep_$n.log)"
done
cd $ROOT
MPDATA_HIP_LIB=$ROOT/codesign-kernels_amd/libmpdata_hip_stamps.so python3 tools/wave_timeline.py --out $OUT/wave_timeline.json > $OUT/wave_timeline.log 2>&1
tail -8 $OUT/wave_timeline.log
python3 tools/closure_summary.py $OUT
