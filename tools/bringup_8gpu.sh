#!/bin/bash
# Bring-up of the multi-GPU path on a node with N >= 2 MI355X (nothing here has ever run on more than one
# physical GPU: DESIGN.md 6).  One go, in the order that localises a failure:
#   1. the -m gpu tests that need two or more devices (tests/test_multi_real_devices.py): sharded plans on
#      [0,1] and on all devices, transports rccl | p2p | direct, host- and root-GPU-origin arrays, run_uw,
#      the Fortran driver's mode = device, bench.py --gpus 2 with the plain command; a case that fails in the
#      default (queued) ordering is retried under MPDATA_MULTI_SYNC=1 and the report says which mode passed
#      (gpurun_out/multi_real_devices.json);
#   2. the Fortran driver at BASELINE.json configs[4]: ncrms = 65536 x N, 25 tracers, mode = device
#      (global arrays generated on the root GPU; 13.45 GB of f per GPU) -- ranks seen, scatter / gather
#      seconds and GB/s per link, HIP kernel seconds;
#   3. the scaling curve: bench.py --gpus 1, 2, 4, ..., N with the plain command (bench.py starts its ranks).
# Usage: tools/bringup_8gpu.sh [N]        (N: default = all devices)   output: gpurun_out/bringup/
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/bringup
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$ROOT"
NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())")
N=${1:-$NDEV}
echo "devices on this node: $NDEV, using $N" | tee "$OUT/summary.txt"
if [ "$N" -lt 2 ]; then echo "needs two or more GPUs" | tee -a "$OUT/summary.txt"; exit 2; fi

echo "== 1. tests/test_multi_real_devices.py" | tee -a "$OUT/summary.txt"
timeout -k 10 3000 python3 -m pytest tests/test_multi_real_devices.py -m gpu -q -x > "$OUT/1_tests.log" 2>&1
rc1=$?
tail -3 "$OUT/1_tests.log" | tee -a "$OUT/summary.txt"

echo "== 2. Fortran driver, configs[4]: ncrms = 65536 x $N, 25 tracers, mode = device" | tee -a "$OUT/summary.txt"
rc2=0
for SYNC in 0 1; do
  if [ $SYNC = 1 ]; then export MPDATA_MULTI_SYNC=1; echo "   (retry under MPDATA_MULTI_SYNC=1)" | tee -a "$OUT/summary.txt"; fi
  timeout -k 10 1500 codesign-kernels_amd/fortran/advect $((65536 * N)) 32 28 1 1 - - 25 "$N" device \
      > "$OUT/2_fortran_sync$SYNC.log" 2>&1
  rc2=$?
  grep -E "mode:|RCCL ranks|scatter|gather|HIP|checksum|Gcu|cell" "$OUT/2_fortran_sync$SYNC.log" | tee -a "$OUT/summary.txt"
  [ $rc2 = 0 ] && break
done
unset MPDATA_MULTI_SYNC

echo "== 3. bench.py --gpus 1 .. $N (plain command)" | tee -a "$OUT/summary.txt"
rc3=0
G=1
while [ "$G" -le "$N" ]; do
  timeout -k 10 1500 python3 bench.py --gpus "$G" --steps 20 --warmup 5 > "$OUT/3_bench_gpus$G.json" 2> "$OUT/3_bench_gpus$G.err"
  rc=$?
  [ $rc != 0 ] && rc3=$rc
  python3 - "$OUT/3_bench_gpus$G.json" "$G" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    tb = d.get("tracer_batched", {})
    sg = d.get("scatter_gather", {})
    print("   gpus %s: value %.4g cu/s  ms/step %.4f  frac %.3f  ranks_seen %s  | 25 tracers: %s cu/s | scatter %s GB/s per link, gather %s"
          % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["ranks_seen"],
             ("%.4g" % tb["value"]) if "value" in tb else tb.get("error"),
             ("%.1f" % sg["scatter_GBs_per_link"]) if "scatter_GBs_per_link" in sg else sg.get("error", "-"),
             ("%.1f" % sg["gather_GBs_per_link"]) if "gather_GBs_per_link" in sg else "-"))
except Exception as exc:
    print("   gpus %s: no line (%r)" % (sys.argv[2], exc))
PY
  G=$((G * 2))
done
echo "rc: tests $rc1, fortran $rc2, bench $rc3" | tee -a "$OUT/summary.txt"
[ $rc1 = 0 ] && [ $rc2 = 0 ] && [ $rc3 = 0 ]
