#!/usr/bin/env bash
# Same-box A/B of environment switches: scripts/ab_env.sh <rounds> "VAR=a" "VAR=b" ...  (interleaved bench.py runs, ms per step)
rounds=$1; shift
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for r in $(seq $rounds); do
  for v in "$@"; do
    ms=$(env $v python bench.py --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} 2>/dev/null | grep -o 'ms_per_step": [0-9.]*' | cut -d' ' -f2)
    echo "round $r  $v  $ms"
  done
done
