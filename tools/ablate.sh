# timing ablations of the x-march kernel (results of the ablated builds are WRONG by design)
set -e
mkdir -p gpurun_out; : > gpurun_out/ablate.log
for i in 1 2; do
  for v in "" _nocomp _nomem _nodma; do
    echo "== lib$v $i" >> gpurun_out/ablate.log
    MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$v.so timeout -k 10 120 python bench.py --steps 100 --warmup 100 --no-cpu-baseline --no-batched --no-fp32 --no-bwk >> gpurun_out/ablate.log 2>&1
    echo "== lib$v T25 $i" >> gpurun_out/ablate.log
    MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$v.so timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batched --no-fp32 --no-bwk --tracers 25 --ncrms-per-gpu 65536 >> gpurun_out/ablate.log 2>&1
  done
done
grep -E "^==|ms_per_step" gpurun_out/ablate.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('=='): print(l.strip(), end=' ')
    else:
        d=json.loads(l); print('%.3f ms  %.1f Gcu/s'%(d['ms_per_step'], d['value']/1e9))
"
