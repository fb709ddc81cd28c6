#!/usr/bin/env bash
# One development pass on the GPU box: build, selected tests, recurrence timelines, bench.  Outputs under gpurun_out/<tag>/.
#   scripts/gpu_iter.sh <tag> "<pytest args or ''>" [steps: lstm bench prof ...]
tag=${1:-it}; pyt=${2:-}; shift 2 || true
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
if [ -n "$pyt" ]; then
  timeout 1800 python -m pytest $pyt -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
  tail -12 $out/pytest.log
fi
for step in "$@"; do
  case $step in
    lstm) TIMELINE=1 timeout 600 python scripts/bench_lstm_step.py cfg2 > $out/lstm_step_cfg2.txt 2>&1; grep -E "persistent|phase durations|launch per" $out/lstm_step_cfg2.txt | cut -c1-600;;
    lstm5) TIMELINE=1 timeout 600 python scripts/bench_lstm_step.py cfg5 > $out/lstm_step_cfg5.txt 2>&1; grep -E "persistent (fwd|bwd):" $out/lstm_step_cfg5.txt;;
    dec) timeout 600 python scripts/bench_lstm_decoder.py > $out/lstm_decoder.txt 2>&1; tail -5 $out/lstm_decoder.txt;;
    bench) timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads(open('$out/bench.json').read().strip().split('\n')[-1])
print('ms/step', d['ms_per_step'], 'utt/s', d['value'], 'rec', d['recurrence'].get('lstm_fwd_us_per_step'), d['recurrence'].get('lstm_bwd_us_per_step'), 'roof', d['roofline'] and d['roofline']['frac'])
PY
    ;;
    bench4|bench5|bench3) c=cfg${step#bench}; timeout 900 python bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_$c.json 2> $out/bench_$c.err; head -c 300 $out/bench_$c.json; echo;;
    prof) bash scripts/prof_cfg.sh cfg2 200 > $out/prof_cfg2.log 2>&1; cp gpurun_out/prof_cfg2/summary.txt $out/cfg2_summary.txt; cp gpurun_out/prof_cfg2/timeline.txt $out/cfg2_timeline.txt; cp gpurun_out/prof_cfg2/b_kernel_stats.csv $out/cfg2_kernel_stats.csv; head -20 $out/cfg2_summary.txt;;
    *) echo "unknown step $step";;
  esac
done
