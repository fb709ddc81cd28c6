#!/usr/bin/env bash
# Baseline probe: recurrence phase timelines + bench, into gpurun_out/<tag>/
tag=${1:-r03a}; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
TIMELINE=1 timeout 600 python scripts/bench_lstm_step.py cfg2 > $out/lstm_step_cfg2.txt 2>&1
TIMELINE=1 timeout 600 python scripts/bench_lstm_step.py cfg5 > $out/lstm_step_cfg5.txt 2>&1
timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
head -c 600 $out/bench.json; echo
cat $out/lstm_step_cfg2.txt
