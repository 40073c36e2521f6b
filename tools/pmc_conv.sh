# usage (on the GPU box): bash tools/pmc_conv.sh <outdir>   -- SQ busy / MFMA-busy counters of the stage-9 conv kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/pmc_conv}
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/a -o a -- python tools/perf_conv.py 9 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/b -o b -- python tools/perf_conv.py 9 3 > /dev/null 2>&1
ls $O/a $O/b
