#!/usr/bin/env bash
# One GPU-box call of round 4: scripts/gpu_call.sh <tag> <step> [<step> ...]; outputs under gpurun_out/<tag>/
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1 || { tail -20 $out/build.log; exit 1; }
for step in "$@"; do
  echo "=== $step"
  case $step in
    calib) timeout 900 python scripts/calib_blas.py all > $out/calib.txt 2>&1; cat $out/calib.txt;;
    calib4) timeout 600 python scripts/calib_blas.py cfg4 > $out/calib4.txt 2>&1; cat $out/calib4.txt;;
    tests) timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -8 $out/pytest.log;;
    bench) timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; head -c 1200 $out/bench.json; echo; tail -3 $out/bench.err;;
    benchq) timeout 900 python bench.py --no-cpu-baseline --no-roofline > $out/benchq.json 2> $out/benchq.err; head -c 400 $out/benchq.json; echo;;
    detail2|detail4|detail5) c=cfg${step#detail}; timeout 900 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --gemm-detail > $out/bench_$c.json 2> $out/detail_$c.txt; head -c 300 $out/bench_$c.json; echo; cat $out/detail_$c.txt | tail -40;;
    bench3|bench4|bench5) c=cfg${step#bench}; timeout 900 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/benchq_$c.json 2> $out/benchq_$c.err; head -c 300 $out/benchq_$c.json; echo;;
    prof2|prof4|prof5) c=cfg${step#prof}; n=30; [ $c = cfg2 ] && n=200; bash scripts/prof_cfg.sh $c $n > $out/prof_$c.log 2>&1; cp gpurun_out/prof_$c/summary.txt $out/${c}_summary.txt; cp gpurun_out/prof_$c/timeline.txt $out/${c}_timeline.txt; cp gpurun_out/prof_$c/b_kernel_stats.csv $out/${c}_kernel_stats.csv; head -22 $out/${c}_summary.txt;;
    py:*) f=${step#py:}; n=$(basename ${f%% *} .py); timeout 900 python $f > $out/$n.txt 2>&1; echo "rc=$?"; tail -40 $out/$n.txt;;
    sh:*) c=${step#sh:}; echo "$c"; timeout 1500 bash -c "$c" 2>&1 | grep -v amdgpu.ids | tail -40;;
    t:*) t=${step#t:}; timeout 1800 python -m pytest $t -x -q -m gpu > $out/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $out/pytest_sel.log; tail -15 $out/pytest_sel.log;;
    *) echo "unknown step $step";;
  esac
done
