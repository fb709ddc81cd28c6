#!/usr/bin/env bash
# One GPU-box pass: tests, bench, rocprofv3 kernel stats.  Outputs under gpurun_out/<tag>/.
#   scripts/gpu_round.sh <tag> [tests|notests] [prof|noprof]
tag=${1:-r02a}; do_tests=${2:-tests}; do_prof=${3:-prof}
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
if [ "$do_tests" = tests ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
  tail -15 $out/pytest.log
fi
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
head -c 1500 $out/bench.json; echo
if [ "$do_prof" = prof ]; then
  root=$PWD
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o b --output-format csv -- python $root/bench.py --no-cpu-baseline > $out/prof_bench.json 2> $out/prof.err)
  d=$(dirname $(find $out/prof -name 'b_kernel_stats.csv' | head -1))
  python scripts/prof_summary.py $d b > $out/summary.txt 2>&1
  cp $d/b_kernel_stats.csv $out/kernel_stats.csv
  head -24 $out/summary.txt
  rm -rf $out/prof
fi
