# Shader clock and VALU busy fraction of the tracer-batch kernel: GRBM_GUI_ACTIVE / kernel time, SQ counters.
# usage (GPU box): bash tools/clock_from_pmc.sh     (MPDATA_WM_TPW1=1 in the environment: one tracer per wave)
ROOT=$PWD; OUT=$ROOT/gpurun_out/clk; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
X="--no-cpu-baseline --no-fp32 --no-bwk --no-reflayout"
for tag in tpw2 tpw1; do
  if [ $tag = tpw1 ]; then export MPDATA_WM_TPW1=1; else unset MPDATA_WM_TPW1; fi
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/${tag}_clk -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --prewarm-ms 0 --batched-steps 4 $X > $OUT/$tag.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/${tag}_sq -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --prewarm-ms 0 --batched-steps 4 $X > $OUT/${tag}_sq.log 2>&1 || exit 1
done
cd $ROOT
python3 - <<'PY'
import csv,glob,collections
for tag in ('tpw2','tpw1'):
    f=glob.glob(f'gpurun_out/clk/{tag}_clk/**/*counter_collection.csv',recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if 'wm_kernel' in r['Kernel_Name'] and 'false' in r['Kernel_Name']:
            dur=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
            print(tag,'dur_us %.0f'%(dur/1e3),'clock %.3f GHz'%(float(r['Counter_Value'])/8/dur))
    f=glob.glob(f'gpurun_out/clk/{tag}_sq/**/*counter_collection.csv',recursive=True)[0]
    acc=collections.defaultdict(float); nd=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if 'wm_kernel' in r['Kernel_Name'] and 'false' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); nd[r['Counter_Name']].add(r['Dispatch_Id'])
    a={k:v/len(nd[k]) for k,v in acc.items()}
    w=a['SQ_WAVES']
    print(tag,{k:round(v/w,1) for k,v in a.items()})
PY
