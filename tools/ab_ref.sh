# usage: bash tools/ab_ref.sh suffixA suffixB ...   ("-" = default library)
# interleaved bench.py runs, reference-layout device call block only (x-march kernel)
F="--steps 40 --warmup 10 --no-cpu-baseline --no-fp32 --no-bwk --no-host-call --no-shared-block --no-fresh-uw --no-x2 --no-batched"
for i in 1 2 3; do for v in "$@"; do
  s=$v; [ "$s" = "-" ] && s=""
  echo "== lib$v $i"
  MPDATA_HIP_LIB=$PWD/codesign-kernels_amd/libmpdata_hip$s.so timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['reference_layout_device_call']; print(round(d['value']/1e9,2), round(d['roofline']['kernel_ms_avg'],4), round(d['roofline']['frac'],4))" || exit 1
done; done
