# usage: bash tools/ab_uw.sh suffixA suffixB ...   ("-" = default library; "x@legacy" = MPDATA_LAYOUT_LEGACY=1)
# interleaved tools/uw_bench.py runs of several library builds (cold plan run, run_uw, conversions)
mkdir -p gpurun_out; : > gpurun_out/ab_uw.log
for i in 1 2 3; do
  for v in "$@"; do
    s=${v%@*}; [ "$s" = "-" ] && s=""
    if [ "$v" != "${v%@legacy}" ]; then export MPDATA_LAYOUT_LEGACY=1; else unset MPDATA_LAYOUT_LEGACY; fi
    echo "== lib$v $i" >> gpurun_out/ab_uw.log
    MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$s.so timeout -k 10 200 python tools/uw_bench.py $UW_ARGS >> gpurun_out/ab_uw.log 2>&1 || exit 1
  done
done
grep -vE "amdgpu.ids" gpurun_out/ab_uw.log
