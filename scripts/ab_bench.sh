#!/usr/bin/env bash
# Same-box A/B of library variants: scripts/ab_bench.sh <rounds> <variant.so> [...]; "cur" = the tree's own build.
# Interleaved rounds of `bench.py --no-cpu-baseline --no-roofline` (ms per step) -- box-to-box spread is +-2 %, so a
# kernel change below that only shows in such a comparison (cdna_hip_programming.md 5.4 rule 24).
rounds=$1; shift
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cp ecog2txt_amd/libecog2txt_hip.so /tmp/cur.so
for r in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = cur ]; then cp /tmp/cur.so ecog2txt_amd/libecog2txt_hip.so; else cp "$v" ecog2txt_amd/libecog2txt_hip.so; fi
    ms=$(python bench.py --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} 2>/dev/null | grep -o 'ms_per_step": [0-9.]*' | cut -d' ' -f2)
    echo "round $r  $(basename $v)  $ms"
  done
done
cp /tmp/cur.so ecog2txt_amd/libecog2txt_hip.so
