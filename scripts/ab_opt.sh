#!/usr/bin/env bash
# Same-box A/B of engine options: scripts/ab_opt.sh <cfg> <rounds> "k=v" "k=v" ...  (interleaved bench.py runs, ms per step)
cfg=$1; rounds=$2; shift; shift
for r in $(seq $rounds); do
  for o in "$@"; do
    ms=$(python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-configs --engine-option $o 2>/dev/null | grep -o 'ms_per_step": [0-9.]*' | cut -d' ' -f2)
    echo "$cfg round $r  $o  $ms"
  done
done
