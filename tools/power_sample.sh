#!/bin/bash
# Socket power and clocks (rocm-smi) while the headline kernel / the 25-tracer batch / the bare streaming kernel loop
# for a few seconds each: is the chip at its power cap while these kernels run?  -> gpurun_out/power_sample.txt
OUT=${1:-gpurun_out/power_sample.txt}
sample() {  # $1 = label; samples until the file /tmp/ps_stop appears
  while [ ! -e /tmp/ps_stop ]; do
    echo "== $1 $(date +%s.%N)"
    rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -iE "power|sclk|mclk|fclk" | head -8
    sleep 0.25
  done
}
run() {  # label, python args...
  rm -f /tmp/ps_stop
  sample "$1" >> $OUT &
  SP=$!
  shift
  "$@" 2>&1 | grep -v amdgpu.ids | tail -3 >> $OUT
  touch /tmp/ps_stop; wait $SP
}
: > $OUT
rocm-smi --showmaxpower --showpower 2>/dev/null | head -12 >> $OUT
run idle sleep 1.5
run headline_1_tracer python3 tools/uw_bench.py --no-uw --no-conv --steps 6000 --sets 12
if [ -e codesign-kernels_amd/libmpdata_hip_firstpass.so ]; then
  MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip_firstpass.so run upwind_pass_only python3 tools/uw_bench.py --no-uw --no-conv --steps 6000 --sets 12
fi
if [ -e codesign-kernels_amd/libmpdata_hip_nocomp.so ]; then
  MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip_nocomp.so run data_movement_only python3 tools/uw_bench.py --no-uw --no-conv --steps 6000 --sets 12
fi
run linear_stream_3r1w python3 -c "
import codesign_kernels_amd as M
for i in range(6): print('stream 3R:1W nontemporal GB/s', M.stream_ceiling(iters=1500))"
run batch_25_tracers python3 tools/uw_bench.py --no-plan --no-uw --no-conv --sets 4 --batch 25 --batch-steps 400
rm -f /tmp/ps_stop
