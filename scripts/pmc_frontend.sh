#!/usr/bin/env bash
# HBM bytes of cfg5's front-end kernels per step, both input forms: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes,
# --kernel-trace only alongside) over eager train steps of bench.py; FETCH_SIZE x 2 (gfx950), KiB units (MI355X_MICROARCH.md).
#   scripts/pmc_frontend.sh -> gpurun_out/pmc_frontend/summary.txt
root=$PWD; out=$PWD/gpurun_out/pmc_frontend; mkdir -p $out
export TMPDIR=/tmp
for form in fp32 bf16; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $out/p_${form}_$ctr -o p --output-format csv -- python $root/bench.py --config cfg5 --inputs $form --steps 3 --warmup 1 --no-graph --no-roofline --no-cpu-baseline --no-configs > $out/${form}_$ctr.log 2>&1)
    f=$(find $out/p_${form}_$ctr -name 'p_counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $out/${form}_$ctr.csv
    rm -rf $out/p_${form}_$ctr
  done
done
python - <<'PY' > $out/summary.txt
import csv
out = 'gpurun_out/pmc_frontend'
NSTEP = 4                                    # 1 warm-up + 3 timed eager steps per pass
def launches(form, ctr, pred):
    rows = list(csv.DictReader(open('%s/%s_%s.csv' % (out, form, ctr))))
    v = [(int(r['Dispatch_Id']), float(r['Counter_Value']) * 1024.0 * (2.0 if ctr == 'FETCH_SIZE' else 1.0))
         for r in rows if pred(r['Kernel_Name'], int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1))]
    return [x for _, x in sorted(v)]
for form in ('fp32', 'bf16'):
    print('cfg5, inputs %s: HBM bytes per launch of the forward front-end kernels (rocprofv3 --pmc, FETCH_SIZE x 2 / WRITE_SIZE)' % form)
    tot = 0.0
    kinds = [('k_seq_lengths_tail_f32', lambda n, g: 'k_seq_lengths_tail' in n, None),
             ('k_conv_fwd_ws<4> (one pass over fp32 x; training form: writes the packed copy)', lambda n, g: 'k_conv_fwd' in n, None),
             ('k_conv_pack', lambda n, g: 'k_conv_pack' in n, None),
             # the 334-workgroup launches of the K-contiguous 128 x 128 instance: per step the conv product first (bf16 form only), then
             # two small products of the auxiliary head with the same grid
             ('conv product on the packed rows (k_gemm_nt<128,128>, 42752 x 100 x 12352)', lambda n, g: 'k_gemm_nt<128, 128, 2, 2, true, false' in n and g == 334, 0)]
    for name, pred, pick in kinds:
        rd, wr = launches(form, 'FETCH_SIZE', pred), launches(form, 'WRITE_SIZE', pred)
        if pick is not None:
            if form != 'bf16' or not rd:
                continue
            per = len(rd) // NSTEP
            rd, wr = rd[pick::per], wr[pick::per]
        if not rd:
            continue
        once = len(rd) < NSTEP                # issued once per fit (staging), not per step
        r_, w_ = sum(rd) / len(rd), sum(wr) / len(wr)
        print('  %-82s %s  read %7.1f MB  written %7.1f MB' % (name, 'once per fit' if once else 'per step    ', r_ / 1e6, w_ / 1e6))
        if not once:
            tot += r_ + w_
    print('  forward front-end per step: %.2f GB; with the 1.06 GB the conv weight gradient reads from the packed rows (both forms): %.2f GB' % (tot / 1e9, tot / 1e9 + 1.056))
PY
cat $out/summary.txt
