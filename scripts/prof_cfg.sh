# rocprofv3 kernel trace of bench.py for one config: stats summary + timeline of one step -> gpurun_out/prof_<cfg>/
cfg=$1; steps=${2:-30}
root=$PWD; out=$PWD/gpurun_out/prof_$cfg; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $out/p -o b --output-format csv -- python $root/bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-roofline --no-configs $BENCH_ARGS > $out/bench.json 2> $out/err.txt
cd $root; d=$(dirname $(find $out/p -name 'b_kernel_stats.csv' | head -1)); python scripts/prof_summary.py $d b > $out/summary.txt 2>&1; cp $d/b_kernel_stats.csv $out/
python scripts/step_timeline.py $d/b_kernel_trace.csv ${TL_MIN:-8} > $out/timeline.txt 2>&1
rm -rf $out/p; head -16 $out/summary.txt; cat $out/timeline.txt; cat $out/bench.json | head -c 300
