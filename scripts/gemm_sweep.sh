#!/usr/bin/env bash
# bench.py --gemm-detail under a list of "ENV=VAL ..." settings; one summary line + the K-major rows each
out=gpurun_out/${1:-sweep}; mkdir -p $out; shift
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== $v"
  env $v timeout 300 python bench.py --no-cpu-baseline --gemm-detail 2> $out/detail_$i.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:(round(v['us_per_step'],1), round(v['frac'],3)) for k,v in d['roofline_all_gemm_instances'].items()})"
  grep -E "^(tn|nt)" $out/detail_$i.txt | awk '{print}' | head -40
done
