F="--dtype f32 --steps 60 --warmup 20 --no-cpu-baseline --no-fp32 --no-bwk --no-reflayout --no-host-call --no-shared-block --no-fresh-uw --no-x2 --no-batched"
for i in 1 2 3; do for s in "" _x32; do
  echo "== lib$s $i"
  MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$s.so timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['roofline']['kernel_ms_avg'], d['roofline']['frac'])" || exit 1
done; done
