# EXACT reference-layout device call, interleaved: register park (default) / park array + finishing kernel
# (MPDATA_EXACT_FLUX=hbm, round 4) / no park (MPDATA_EXACT_FLUX=sum) / FAST.  -> gpurun_out/ab_exact_call.log
mkdir -p gpurun_out; L=gpurun_out/ab_exact_call.log; : > $L
for i in 1 2 3; do
  for v in regs hbm sum; do
    if [ $v = regs ]; then unset MPDATA_EXACT_FLUX; else export MPDATA_EXACT_FLUX=$v; fi
    echo "== exact flux=$v pass $i" >> $L
    timeout -k 10 300 python tools/wm_bench.py --variant exact --no-t25 --steps 40 >> $L 2>&1 || exit 1
  done
  unset MPDATA_EXACT_FLUX
  echo "== fast pass $i" >> $L
  timeout -k 10 300 python tools/wm_bench.py --variant fast --no-t25 --steps 40 >> $L 2>&1 || exit 1
done
grep -E "^==|x-march" $L
