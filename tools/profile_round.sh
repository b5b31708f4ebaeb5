#!/bin/bash
# Round profile: kernel-trace stats + three separate PMC passes of the headline bench, then the
# summaries the judge reads are written into profiles/ by tools/pmc_summary.py.
#   usage (on the GPU box):  bash tools/profile_round.sh r01
# rocprofv3 rules of this pool: the program itself after `--`; --pmc never together with
# --kernel-trace/--stats; one counter group per pass.
set -e
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-batched --no-fp32 --no-bwk"
S="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batched --no-fp32 --no-bwk"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o run -- $B > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- $S > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/write -o run -- $S > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/sq -o run -- $S > $OUT/sq.log 2>&1
# second kernel (biharmonic_wk_scalar, nelemd=5400): kernel trace + HBM traffic
W="python3 $ROOT/tools/bwk_bench.py --child -"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bwk_kt -o run -- $W > $OUT/bwk_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bwk_fetch -o run -- $W > $OUT/bwk_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bwk_write -o run -- $W > $OUT/bwk_write.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $TAG
