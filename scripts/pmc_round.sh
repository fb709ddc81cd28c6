# PMC passes (separate runs, --kernel-trace only alongside) over scripts/roofline_kernels.py -> gpurun_out/pmc_<cfg>/*.csv
cfg=${1:-cfg2}
root=$PWD; out=$PWD/gpurun_out/pmc_$cfg; mkdir -p $out
export TMPDIR=/tmp
run() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $out/$name -o p --output-format csv -- python $root/scripts/roofline_kernels.py $cfg 3 > $out/$name.log 2>&1); f=$(find $out/$name -name 'p_counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $out/$name.csv; rm -rf $out/$name; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
ls -la $out; grep PRODUCTS $out/sq.log
