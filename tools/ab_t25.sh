# usage: bash tools/ab_t25.sh suffixA suffixB ...   ("-" = default library; a suffix ending in "@1" sets MPDATA_WM_TPW1=1)
# interleaved 25-tracer plan runs of several library builds
mkdir -p gpurun_out; : > gpurun_out/ab_t25.log
for i in 1 2 3; do
  for v in "$@"; do
    s=${v%@*}; [ "$s" = "-" ] && s=""
    if [ "$v" != "${v%@1}" ]; then export MPDATA_WM_TPW1=1; else unset MPDATA_WM_TPW1; fi
    echo "== lib$v $i" >> gpurun_out/ab_t25.log
    MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$s.so timeout -k 10 200 python tools/wm_bench.py --no-ref --no-t1 --t25-steps 10 >> gpurun_out/ab_t25.log 2>&1 || exit 1
  done
done
grep -E "^==|T=25" gpurun_out/ab_t25.log
