#!/bin/bash
# Round profile (on the GPU box): kernel stats of a stage-9 and a stage-5 iteration, the bench command's kernel trace, and
# the PMC traffic / busy passes of the conv family.  usage: bash tools/profile_round.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/prof}
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s9 -o s9 -- python3 bench.py --stages 9 --steps 3 --warmup 1 --no-cpu-baseline > $O/s9.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s5 -o s5 -- python3 bench.py --stages 5 --steps 5 --warmup 1 --no-cpu-baseline --graph-stages none > $O/s5.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 tools/perf_conv.py 9 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 tools/perf_conv.py 9 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -o a -- python3 tools/perf_conv.py 9 3 > /dev/null 2>&1
find $O -name "*.csv" | head -30
