# usage: bash ab_wm.sh suffixA suffixB ... ; interleaved wm_bench runs (T=1 and T=25)
mkdir -p gpurun_out; : > gpurun_out/ab_wm.log
for v in "$@"; do
  s=$v; [ "$v" = "-" ] && s=""
  MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$s.so timeout -k 10 600 python -m pytest tests/test_plan_wavemajor.py -m gpu -x -q -k "golden or shapes" > gpurun_out/ab_wm_tests$s.log 2>&1 || { echo "PARITY FAIL $v"; tail -15 gpurun_out/ab_wm_tests$s.log; }
  tail -1 gpurun_out/ab_wm_tests$s.log
done
for i in 1 2 3; do
  for v in "$@"; do
    s=$v; [ "$v" = "-" ] && s=""
    echo "== lib$s $i" >> gpurun_out/ab_wm.log
    MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$s.so timeout -k 10 200 python tools/wm_bench.py --no-ref >> gpurun_out/ab_wm.log 2>&1 || exit 1
  done
done
grep -E "^==|plan wave" gpurun_out/ab_wm.log
