#!/bin/bash
# Round profile: kernel-trace stats + separate PMC passes of the headline bench (1 tracer, plan
# API, wave-major layout) and of the 25-tracer batch, then tools/pmc_summary.py writes the
# summaries the judge reads into profiles/.
#   usage (on the GPU box):  bash tools/profile_round.sh r04   (tools/record_round.sh runs it with the other records of a round)
# The wake-up launches of bench.py cycle through its scratch field sets (round 4): every dispatch of the kernel trace is a
# cold one, the CSV's plain average is the cold figure.
# rocprofv3 rules of this pool: the program itself after `--`; --pmc never together with
# --kernel-trace/--stats; one counter group per pass.
set -e
TAG=${1:-r05}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# every pass runs ONLY the block whose kernel it profiles (the side blocks launch kernels of the same
# name at other sizes -- the chunked host call did, round 2 -- and tools/pmc_summary.py additionally keeps
# only the full-size dispatches of a kernel)
X="--no-cpu-baseline --no-fp32 --no-bwk --no-exact --no-reflayout --no-host-call --no-shared-block --no-fresh-uw --no-x2"
B="python3 $ROOT/bench.py --steps 20 --warmup 5 $X --no-batched"
S="python3 $ROOT/bench.py --steps 3 --warmup 1 --prewarm-ms 0 $X --no-batched"
T="python3 $ROOT/bench.py --steps 2 --warmup 1 --prewarm-ms 0 --batched-steps 2 $X"
R="python3 $ROOT/bench.py --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline --no-fp32 --no-bwk --no-exact --no-batched --no-host-call --no-shared-block --no-fresh-uw --no-x2"
U="python3 $ROOT/bench.py --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline --no-fp32 --no-bwk --no-exact --no-batched --no-host-call --no-shared-block --no-reflayout"
N="python3 $ROOT/tools/nlk_bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o run -- $B > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- $S > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/write -o run -- $S > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/sq -o run -- $S > $OUT/sq.log 2>&1
# 25 tracers: kernel trace, HBM traffic, L2 hit rate, SQ
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t25_kt -o run -- python3 $ROOT/bench.py --steps 20 --warmup 2 $X > $OUT/t25_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/t25_fetch -o run -- $T > $OUT/t25_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/t25_write -o run -- $T > $OUT/t25_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/t25_tcc -o run -- $T > $OUT/t25_tcc.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/t25_sq -o run -- $T > $OUT/t25_sq.log 2>&1
# reference-layout device call (x-march kernel): HBM traffic
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ref_fetch -o run -- $R > $OUT/ref_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/ref_write -o run -- $R > $OUT/ref_write.log 2>&1
# mpdata_plan_run_uw (u, w from the reference layout): kernel trace of the steady block, HBM traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/uw_kt -o run -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-bwk --no-exact --no-batched --no-host-call --no-shared-block --no-reflayout > $OUT/uw_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/uw_fetch -o run -- $U > $OUT/uw_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/uw_write -o run -- $U > $OUT/uw_write.log 2>&1
# third kernel (high-order flux nest) on the 32 x mesh, local and random connectivity: kernel trace + HBM traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nlk_kt -o run -- $N > $OUT/nlk_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/nlk_fetch -o run -- $N > $OUT/nlk_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/nlk_write -o run -- $N > $OUT/nlk_write.log 2>&1
# second kernel (biharmonic_wk_scalar, nelemd=5400): kernel trace + HBM traffic
W="python3 $ROOT/tools/bwk_bench.py --child -"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bwk_kt -o run -- $W > $OUT/bwk_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bwk_fetch -o run -- $W > $OUT/bwk_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bwk_write -o run -- $W > $OUT/bwk_write.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $TAG
