# Round-end evidence: tests, bench (cfg2 with roofline + CPU baseline), rocprofv3 stats + timeline, the other configs.
tag=${1:-r03}
out=$PWD/gpurun_out/$tag; mkdir -p $out
bash scripts/gpu_round.sh $tag tests noprof > $out/round.log 2>&1
tail -5 $out/pytest.log
bash scripts/prof_cfg.sh cfg2 200 > $out/prof_cfg2.log 2>&1
cp gpurun_out/prof_cfg2/summary.txt $out/cfg2_summary.txt; cp gpurun_out/prof_cfg2/timeline.txt $out/cfg2_timeline.txt; cp gpurun_out/prof_cfg2/b_kernel_stats.csv $out/cfg2_kernel_stats.csv; cp gpurun_out/prof_cfg2/bench.json $out/cfg2_prof_bench.json
for c in cfg3 cfg4 cfg5; do timeout 600 python bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err; head -c 200 $out/bench_$c.json; echo; done
bash scripts/prof_cfg.sh cfg4 30 > $out/prof_cfg4.log 2>&1; cp gpurun_out/prof_cfg4/summary.txt $out/cfg4_summary.txt; cp gpurun_out/prof_cfg4/b_kernel_stats.csv $out/cfg4_kernel_stats.csv
bash scripts/prof_cfg.sh cfg5 30 > $out/prof_cfg5.log 2>&1; cp gpurun_out/prof_cfg5/summary.txt $out/cfg5_summary.txt; cp gpurun_out/prof_cfg5/b_kernel_stats.csv $out/cfg5_kernel_stats.csv
python scripts/bench_conv_front.py cfg5 > $out/conv_front_cfg5.txt 2>&1
python scripts/bench_lstm_big.py > $out/lstm_big.txt 2>&1
python scripts/bench_fit.py 100 > $out/fit.txt 2>&1; tail -3 $out/fit.txt
head -c 600 $out/bench.json
