#!/usr/bin/env bash
# Same-box A/B of DEBUG-library switches (csrc/common.h e2t_dbg_*): scripts/ab_dbg.sh <cfg> <rounds> "VAR=a" "VAR=b" ...
cfg=$1; rounds=$2; shift; shift
(cd scripts && python -c "import _dbg") > /dev/null 2>&1
for r in $(seq $rounds); do
  for v in "$@"; do
    ms=$(env E2T_DEBUG_LIB=1 $v python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-configs 2>/dev/null | grep -o 'ms_per_step": [0-9.]*' | cut -d' ' -f2)
    echo "$cfg round $r  $v  $ms"
  done
done
