# rocprofv3 kernel trace of the data-parallel schedule on one GPU (scripts/dp_step_one_gpu.py): timeline of one step -> gpurun_out/prof_dp_<mode>/
mode=${1:-graph_per_stage}; cfg=${2:-cfg2}
root=$PWD; out=$PWD/gpurun_out/prof_dp_$mode; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && DP_MODES=$mode timeout 600 rocprofv3 --kernel-trace --stats -d $out/p -o b --output-format csv -- python $root/scripts/dp_step_one_gpu.py $cfg 30 > $out/run.txt 2> $out/err.txt
cd $root; d=$(dirname $(find $out/p -name 'b_kernel_stats.csv' | head -1))
python scripts/step_timeline.py $d/b_kernel_trace.csv ${TL_MIN:-4} > $out/timeline.txt 2>&1
rm -rf $out/p; grep "ms per step" $out/run.txt; cat $out/timeline.txt
