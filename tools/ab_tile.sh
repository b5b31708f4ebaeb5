#!/bin/bash
# usage: bash tools/ab_tile.sh tileA tileB ...   (-1 = the default choice)
# interleaved bench.py runs, reference-layout device call block only (x-march kernel), with the kernel tiling forced
F="--steps 40 --warmup 10 --no-cpu-baseline --no-fp32 --no-bwk --no-host-call --no-shared-block --no-fresh-uw --no-x2 --no-batched"
for i in 1 2 3; do for t in "$@"; do
  echo -n "pass $i tile $t: "
  timeout -k 10 200 python bench.py $F --tile $t 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['reference_layout_device_call']; print('%.2f Gcu/s  kernel %.4f ms  frac %.4f' % (d['value']/1e9, d['roofline']['kernel_ms_avg'], d['roofline']['frac']))" || exit 1
done; done
