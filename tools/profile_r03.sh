#!/bin/bash
# End-of-round-3 artefacts (on the GPU box): bench lines of the four configs (video with the CPU leg), the bench command under
# rocprofv3 --kernel-trace --stats, kernel stats of stage 9 / 8 / 7 / 5 iterations, PMC passes (HBM traffic, matrix-pipe busy) of
# the two roofline kernel families at stage 9.  usage: bash tools/profile_r03.sh <outdir>  (copy what is judged into profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/final_r03}
mkdir -p $O
python3 bench.py > $O/bench_video.log 2>&1; echo video >> $O/progress
python3 bench.py --config image --no-cpu-baseline > $O/bench_image.log 2>&1; echo image >> $O/progress
python3 bench.py --config video8 --no-cpu-baseline > $O/bench_video8.log 2>&1; echo video8 >> $O/progress
python3 bench.py --config baseline --no-cpu-baseline > $O/bench_baseline.log 2>&1; echo baseline >> $O/progress
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -o b -- python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; echo trace >> $O/progress
python3 tools/roofline_from_trace.py $O/b/b_kernel_trace.csv "conv_wino2r_kernel<0" > $O/roofline_trace_conv.txt 2>&1 || true
python3 tools/roofline_from_trace.py $O/b/b_kernel_trace.csv "conv_wgradw2_kernel<3, 1, 1, 16, false>" > $O/roofline_trace_wgrad.txt 2>&1 || true
cp $(find $O/b -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
bash tools/profile_stages.sh $O/stages 9 8 7 5 > $O/stages.txt 2>&1; echo stages >> $O/progress
# PMC: separate passes (never with the trace domains)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fc -o f -- python3 tools/perf_conv.py 9 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/wc -o w -- python3 tools/perf_conv.py 9 3 > /dev/null 2>&1
python3 tools/pmc_summary.py $(find $O/fc -name "*counter_collection.csv") $(find $O/wc -name "*counter_collection.csv") --roofline-json profiles/roofline_traffic.json conv_fwd profiles/r03_final_pmc_traffic_conv_stage9.csv 2 64 13 144 256 > $O/pmc_traffic_conv_stage9.csv 2>$O/pmc_err_c.txt; echo pmc_conv >> $O/progress
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fw -o f -- python3 tools/perf_wgrad_wino.py 3 9 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/ww -o w -- python3 tools/perf_wgrad_wino.py 3 9 > /dev/null 2>&1
python3 tools/pmc_summary.py $(find $O/fw -name "*counter_collection.csv") $(find $O/ww -name "*counter_collection.csv") --roofline-json profiles/roofline_traffic.json weight_gradient profiles/r03_final_pmc_traffic_wgrad_stage9.csv 2 64 13 144 256 > $O/pmc_traffic_wgrad_stage9.csv 2>$O/pmc_err_w.txt; echo pmc_wgrad >> $O/progress
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/ac -o a -- python3 tools/perf_conv.py 9 3 > /dev/null 2>&1
python3 tools/pmc_busy.py $(find $O/ac -name "*counter_collection.csv") > $O/pmc_busy_conv_stage9.csv 2>/dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/aw -o a -- python3 tools/perf_wgrad_wino.py 3 9 > /dev/null 2>&1
python3 tools/pmc_busy.py $(find $O/aw -name "*counter_collection.csv") > $O/pmc_busy_wgrad_stage9.csv 2>/dev/null; echo busy >> $O/progress
(for sh in 13,144,256 7,114,204 7,91,162; do for v in 1 0; do echo "== $sh HPVG_WINO2R=$v"; HPVG_WINO2R=$v HPVG_PERF_SHAPE=$sh python3 tools/perf_wino_variants.py 2>/dev/null; done; done) > $O/perf_wino2r.txt
[ -x tools/mfma_fillers.bin ] && tools/mfma_fillers.bin > $O/mfma_fillers.txt 2>&1
cp profiles/roofline_traffic.json $O/roofline_traffic.json
rm -rf $O/b $O/fc $O/wc $O/fw $O/ww $O/ac $O/aw
ls $O
