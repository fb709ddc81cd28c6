#!/usr/bin/env bash
# Round evidence, phase 1: rocprofv3 kernel stats + one-step timelines of cfg2 / cfg4 / cfg5, PMC passes of the final kernels (cfg2).
#   scripts/gpu_evidence.sh <tag>      -> gpurun_out/<tag>/ (copy what is to be judged into profiles/)
tag=${1:-r03}; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
for c in cfg2 cfg4 cfg5; do
  n=200; [ $c != cfg2 ] && n=30
  bash scripts/prof_cfg.sh $c $n > $out/prof_$c.log 2>&1
  cp gpurun_out/prof_$c/summary.txt $out/${c}_summary.txt; cp gpurun_out/prof_$c/timeline.txt $out/${c}_timeline.txt
  cp gpurun_out/prof_$c/b_kernel_stats.csv $out/${c}_kernel_stats.csv; cp gpurun_out/prof_$c/bench.json $out/${c}_prof_bench.json
done
bash scripts/pmc_round.sh cfg2 > $out/pmc_cfg2.log 2>&1
python scripts/pmc_kernels_summary.py gpurun_out/pmc_cfg2 ${tag}tmp > $out/pmc_summary.txt 2>&1
mv profiles/${tag}tmp_pmc_kernels.json $out/pmc_kernels.json; mv profiles/${tag}tmp_pmc_gemm.json $out/pmc_gemm.json
head -12 $out/cfg2_summary.txt; tail -3 $out/pmc_summary.txt | cut -c1-300
