set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -3 gpurun_out/ab_tests.log
: > gpurun_out/ab_bench.log
for i in 1 2 3; do
  for v in base fused new; do
    for var in fast exact; do
      echo "== $v $var $i" >> gpurun_out/ab_bench.log
      MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip_$v.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-batched --variant $var >> gpurun_out/ab_bench.log 2>&1
    done
  done
done
python - <<'PY'
import json,re
cur=None
res={}
for line in open('gpurun_out/ab_bench.log'):
    if line.startswith('=='): cur=tuple(line.split()[1:3]); continue
    if line.startswith('{'):
        res.setdefault(cur,[]).append(json.loads(line)['value']/1e9)
for k,v in sorted(res.items()): print(k, ['%.1f'%x for x in v])
PY
