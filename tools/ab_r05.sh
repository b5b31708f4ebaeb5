# Round-5 interleaved A/B records (one box, three passes each):
#   (1) 25 tracers, FAST: the odd tracer in one more wave per tile of the batch launch (default) against a launch of its
#       own behind the batch (MPDATA_WM_SPLIT=1, rounds 2-4)
#   (2) EXACT: limited vertical fluxes parked in registers (default) / in HBM + finishing kernel (MPDATA_EXACT_FLUX=hbm,
#       round 4) / not parked (MPDATA_EXACT_FLUX=sum, flux to 1e-13), one tracer and 25
# usage: bash tools/ab_r05.sh     -> gpurun_out/ab_r05.log
mkdir -p gpurun_out; L=gpurun_out/ab_r05.log; : > $L
for i in 1 2 3; do
  for v in default split; do
    echo "== t25 fast odd-tracer=$v pass $i" >> $L
    if [ $v = split ]; then export MPDATA_WM_SPLIT=1; else unset MPDATA_WM_SPLIT; fi
    timeout -k 10 200 python tools/wm_bench.py --no-ref --no-t1 --t25-steps 10 >> $L 2>&1 || exit 1
  done
  unset MPDATA_WM_SPLIT
  for v in regs hbm sum; do
    echo "== exact flux=$v pass $i" >> $L
    if [ $v = regs ]; then unset MPDATA_EXACT_FLUX; else export MPDATA_EXACT_FLUX=$v; fi
    timeout -k 10 300 python tools/wm_bench.py --variant exact --no-ref --steps 40 --t25-steps 5 >> $L 2>&1 || exit 1
  done
  unset MPDATA_EXACT_FLUX
done
grep -E "^==|T=" $L
