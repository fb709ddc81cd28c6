export TMPDIR=/tmp; mkdir -p gpurun_out/r6h
(cd scripts && python -c "import _dbg") > /dev/null 2>&1
for c in cfg2 cfg4; do for k in 0 2 3 4 6 8; do echo "== $c tail alone, E2T_ADAM_PACK_WGS_PER_CU=$k"; E2T_DEBUG_LIB=1 E2T_ADAM_PACK_WGS_PER_CU=$k timeout 300 python scripts/bench_tail.py $c 2>&1 | grep fused; done; done > gpurun_out/r6h/tail.txt 2>&1
cat gpurun_out/r6h/tail.txt
bash scripts/ab_dbg.sh cfg4 2 E2T_ADAM_PACK_WGS_PER_CU=0 E2T_ADAM_PACK_WGS_PER_CU=2 E2T_ADAM_PACK_WGS_PER_CU=3 E2T_ADAM_PACK_WGS_PER_CU=4 E2T_ADAM_PACK_WGS_PER_CU=6 > gpurun_out/r6h/ab_cfg4.txt 2>&1; cat gpurun_out/r6h/ab_cfg4.txt
bash scripts/ab_dbg.sh cfg2 2 E2T_ADAM_PACK_WGS_PER_CU=0 E2T_ADAM_PACK_WGS_PER_CU=2 E2T_ADAM_PACK_WGS_PER_CU=3 E2T_ADAM_PACK_WGS_PER_CU=4 E2T_ADAM_PACK_WGS_PER_CU=6 > gpurun_out/r6h/ab_cfg2.txt 2>&1; cat gpurun_out/r6h/ab_cfg2.txt
timeout 900 python -m pytest tests/test_gpu_fullsize_parity.py -q -m gpu -s -k "cfg3" > gpurun_out/r6h/pytest.log 2>&1; tail -3 gpurun_out/r6h/pytest.log
