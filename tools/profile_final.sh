#!/bin/bash
# End-of-round artefacts (on the GPU box): the four bench lines, the bench command under rocprofv3 --kernel-trace --stats,
# kernel stats of a stage-9 / 8 / 5 iteration, PMC passes (traffic, matrix-pipe busy) of the Winograd weight gradient at
# stage 9.  usage: bash tools/profile_final.sh <outdir>    (copy what is to be judged into profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/final}
mkdir -p $O
set -e
python3 bench.py > $O/bench_video.log 2>&1; echo video >> $O/progress
python3 bench.py --config image --no-cpu-baseline > $O/bench_image.log 2>&1; echo image >> $O/progress
python3 bench.py --config video8 --no-cpu-baseline > $O/bench_video8.log 2>&1; echo video8 >> $O/progress
python3 bench.py --config baseline --no-cpu-baseline > $O/bench_baseline.log 2>&1; echo baseline >> $O/progress
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -o b -- python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; echo trace >> $O/progress
python3 tools/roofline_from_trace.py $O/b/b_kernel_trace.csv "conv_wino2d_kernel<0" > $O/roofline_trace.txt 2>&1 || true
for st in 9 8 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$st -o s$st -- python3 bench.py --stages $st --steps 3 --warmup 1 --no-cpu-baseline > $O/s$st.log 2>&1; echo s$st >> $O/progress
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 tools/perf_wgrad_wino.py 3 9 > /dev/null 2>&1; echo f >> $O/progress
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 tools/perf_wgrad_wino.py 3 9 > /dev/null 2>&1; echo w >> $O/progress
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -o a -- python3 tools/perf_wgrad_wino.py 3 9 > /dev/null 2>&1; echo a >> $O/progress
rm -f $O/b/*_kernel_trace.csv $O/s*/*_kernel_trace.csv   # large; the stats summaries stay
find $O -name "*.csv" | head -40
