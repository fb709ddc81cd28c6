#!/usr/bin/env bash
# Round evidence in one GPU-box call: the full GPU suite, the default bench line (configs block, roofline, CPU baseline), rocprofv3
# kernel stats + one-step timelines of cfg2 / cfg4 / cfg5 (fp32 and bf16-staged inputs), the PMC passes of cfg2's kernels, the
# vendor-BLAS calibration and the data-parallel schedule on one GPU.
#   scripts/gpu_evidence.sh <tag> [notests]   -> gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>_*)
tag=${1:-r06}; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
if [ "$2" != notests ]; then
  timeout 2400 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
fi
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"; head -c 300 $out/bench_line.json; echo
for c in cfg2 cfg4 cfg5; do
  n=200; [ $c != cfg2 ] && n=30
  bash scripts/prof_cfg.sh $c $n > $out/prof_$c.log 2>&1
  cp gpurun_out/prof_$c/summary.txt $out/${c}_summary.txt; cp gpurun_out/prof_$c/timeline.txt $out/${c}_timeline.txt
  cp gpurun_out/prof_$c/b_kernel_stats.csv $out/${c}_kernel_stats.csv
  head -8 $out/${c}_summary.txt
done
BENCH_ARGS="--inputs bf16" bash scripts/prof_cfg.sh cfg5 30 > $out/prof_cfg5_bf16.log 2>&1
cp gpurun_out/prof_cfg5/summary.txt $out/cfg5_bf16_summary.txt; cp gpurun_out/prof_cfg5/timeline.txt $out/cfg5_bf16_timeline.txt
cp gpurun_out/prof_cfg5/b_kernel_stats.csv $out/cfg5_bf16_kernel_stats.csv
bash scripts/pmc_round.sh cfg2 > $out/pmc.log 2>&1
python scripts/pmc_kernels_summary.py gpurun_out/pmc_cfg2 $tag > $out/pmc_summary.log 2>&1; cp profiles/${tag}_pmc_kernels.json profiles/${tag}_pmc_gemm.json $out/; tail -3 $out/pmc_summary.log
timeout 900 python scripts/calib_blas.py all 2>&1 | grep -v amdgpu.ids > $out/calib_blas.txt
timeout 600 python scripts/dp_step_one_gpu.py 2>&1 | grep "ms per step" > $out/dp_step_one_gpu.txt; cat $out/dp_step_one_gpu.txt
timeout 600 python scripts/dp_step_one_gpu.py cfg4 30 2>&1 | grep "ms per step" >> $out/dp_step_one_gpu.txt
# round 5: the fused tail alone, single-utterance decode latency, fit-level throughput, phase stamps of the recurrences
for c in cfg2 cfg4; do timeout 300 python scripts/bench_tail.py $c 2>&1 | grep -v amdgpu.ids >> $out/tail_alone.txt; done; cat $out/tail_alone.txt
for c in cfg2 cfg4; do timeout 300 python scripts/latency_b1.py $c 2>&1 | grep -v amdgpu.ids >> $out/latency_b1.txt; done; cat $out/latency_b1.txt
timeout 600 python scripts/bench_fit.py 100 2>&1 | grep -v amdgpu.ids > $out/fit_throughput.txt; tail -3 $out/fit_throughput.txt
TIMELINE=1 timeout 600 python scripts/bench_lstm_step.py cfg2 2>&1 | grep -v amdgpu.ids > $out/lstm_step_phases.txt; head -30 $out/lstm_step_phases.txt
ls $out
# round 6: cfg5 front-end HBM bytes on the final tree, the recurrences' side-work probe + DEFER experiment, the data-parallel timeline
bash scripts/pmc_frontend.sh > $out/pmc_frontend.log 2>&1; cp gpurun_out/pmc_frontend/summary.txt $out/pmc_cfg5_frontend.txt; cat $out/pmc_cfg5_frontend.txt
timeout 600 python scripts/probe_rec_sidework.py cfg2 quick 2>&1 | grep -v amdgpu.ids > $out/rec_sidework_quick.txt; cat $out/rec_sidework_quick.txt
bash scripts/prof_dp.sh graph_per_stage cfg2 > $out/dp_timeline_graph_per_stage.txt 2>&1
ls $out
