#!/bin/bash
# In-kernel shader clock of the headline kernel, its two ablation builds and the 25-tracer batch, each after >= 2.5 s of
# back-to-back cold launches (tools/inkernel_clock.py).  Needs the three stamped builds:
#   bash tools/build_variants.sh "_stamps -DMPDWM_STAMPS" "_stfirst -DMPDWM_STAMPS -DMPDWM_ABL_FIRSTPASS" \
#                                "_stnocomp -DMPDWM_STAMPS -DMPDWM_ABL_NOCOMPUTE"
# usage (GPU box): bash tools/inkernel_clock.sh   ->  gpurun_out/inkernel_clock/*.json + summary.json
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/inkernel_clock
mkdir -p $OUT
cd $ROOT
L=$ROOT/codesign-kernels_amd
for pass in 1 2; do
  MPDATA_HIP_LIB=$L/libmpdata_hip_stamps.so   timeout -k 10 300 python3 tools/inkernel_clock.py --tracers 1  --out $OUT/full_t1_$pass.json
  MPDATA_HIP_LIB=$L/libmpdata_hip_stfirst.so  timeout -k 10 300 python3 tools/inkernel_clock.py --tracers 1  --out $OUT/firstpass_t1_$pass.json
  MPDATA_HIP_LIB=$L/libmpdata_hip_stnocomp.so timeout -k 10 300 python3 tools/inkernel_clock.py --tracers 1  --out $OUT/nocompute_t1_$pass.json
  MPDATA_HIP_LIB=$L/libmpdata_hip_stamps.so   timeout -k 10 300 python3 tools/inkernel_clock.py --tracers 25 --out $OUT/full_t25_$pass.json
done
python3 - $OUT <<'PY'
import glob, json, os, sys
out = {"what": "in-kernel shader clock (d s_memtime / d s_memrealtime x 100 MHz, median over the waves of one launch) after >= 2.5 s "
               "of back-to-back cold launches, MI355X, ncrms = 65536, nx = 32, nz = 28, FAST; two interleaved passes",
       "kernels": {}}
for path in sorted(glob.glob(os.path.join(sys.argv[1], "*_[12].json"))):
    name = os.path.basename(path)[:-7]
    d = json.load(open(path))
    ms = [v for k, v in d.items() if k.startswith("ms_per_plan_run")][0]
    out["kernels"].setdefault(name, []).append({"clock_GHz_median": round(d["shader_clock_GHz"]["median"], 4),
                                                "p05": round(d["shader_clock_GHz"]["p05"], 4), "p95": round(d["shader_clock_GHz"]["p95"], 4),
                                                "ms_per_plan_run": round(ms, 4), "waves": d["waves_stamped"], "build": d["build"]})
json.dump(out, open(os.path.join(sys.argv[1], "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
