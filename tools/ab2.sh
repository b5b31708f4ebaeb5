# usage: bash tools/ab2.sh suffixA suffixB ...   (interleaved bench of library builds, FAST variant; "-" = default lib)
set -e
mkdir -p gpurun_out; : > gpurun_out/ab2.log
for v in "$@"; do
  s=$v; [ "$v" = "-" ] && s=""
  MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$s.so timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden or tiles or ragged" > gpurun_out/ab2_tests$s.log 2>&1 || { echo "PARITY FAIL $v"; tail -15 gpurun_out/ab2_tests$s.log; }
done
for i in 1 2 3; do
  for v in "$@"; do
    s=$v; [ "$v" = "-" ] && s=""
    echo "== lib$s $i" >> gpurun_out/ab2.log
    MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$s.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-batched --no-fp32 --no-bwk >> gpurun_out/ab2.log 2>&1
    echo "== lib$s T25 $i" >> gpurun_out/ab2.log
    MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$s.so timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batched --no-fp32 --no-bwk --tracers 25 >> gpurun_out/ab2.log 2>&1
  done
done
python - <<'PY'
import json
cur=None; res={}
for line in open('gpurun_out/ab2.log'):
    if line.startswith('=='): cur=' '.join(line.split()[1:-1]); continue
    if line.startswith('{'): res.setdefault(cur,[]).append(json.loads(line)['value']/1e9)
for k,v in res.items(): print('%-24s'%k, ' '.join('%.1f'%x for x in v))
PY
