"""Summarise the PMC passes of scripts/pmc_round.sh: per kernel (launch averages over the isolated launches of
scripts/roofline_kernels.py) -> profiles/<tag>_pmc_kernels.json and <tag>_pmc_gemm.json (the `roofline.traffic` of bench.py).
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md)."""
import csv, json, sys, collections
d, tag = sys.argv[1], sys.argv[2]
def short(name):
    if 'k_gemm_tn_group' in name: return 'tn128g'
    if 'k_splitk_reduce_group' in name: return 'k_splitk_reduce_group'
    if 'k_gemm_nt<128, 128, 2, 2, true, true' in name: return 'tn128'
    if 'k_gemm_nt<128, 128, 2, 2, true, false' in name: return 'nt128'
    if 'k_gemm_nt<256, 256, 2, 4, false, true' in name: return 'tn256'
    if 'k_gemm_nt<256' in name: return 'nt256'
    for k in ('k_conv_fwd_ws', 'k_conv_fwd', 'k_conv_pack', 'k_splitk_reduce', 'k_lstm_seq_fwd_persist_wide', 'k_lstm_seq_fwd_persist', 'k_lstm_seq_bwd_persist<25', 'k_lstm_seq_bwd_persist',
              'k_lstm_seq_fwd_big', 'k_lstm_seq_bwd_big', 'k_lstm_step_fwd', 'k_lstm_step_bwd'):
        if k in name: return k.replace('<25', '_wide')
    return None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)          # kernel -> {dispatch id: ns} from the SQ pass
for f in ('sq', 'fetch', 'write', 'l2', 'grbm'):
    try:
        rows = list(csv.DictReader(open('%s/%s.csv' % (d, f))))
    except OSError:
        continue
    for r in rows:
        k = short(r['Kernel_Name'])
        if k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if f == 'sq':
                dur[k][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
out = {}
for k, c in acc.items():
    n = len(next(iter(c.values())))
    # every k_gemm instance is launched 3 x (its launches per step): the first pass of the log is the eager step itself, whose
    # launches carry the same names -- all of them are averaged
    o = {name: sum(v) / len(v) for name, v in c.items()}
    e = dict(launches_sampled=n)
    if 'SQ_WAVE_CYCLES' in o:
        wc = o['SQ_WAVE_CYCLES']
        e.update(wave_cycles=round(wc), frac_wait_any=round(o['SQ_WAIT_ANY'] / wc, 3), frac_wait_inst=round(o['SQ_WAIT_INST_ANY'] / wc, 3),
                 frac_active_inst=round(o['SQ_ACTIVE_INST_ANY'] / wc, 3), mfma_busy_cycles=round(o['SQ_VALU_MFMA_BUSY_CYCLES']),
                 sq_busy_cycles=round(o['SQ_BUSY_CYCLES']), lds_bank_conflict_cycles=round(o['SQ_LDS_BANK_CONFLICT']), lds_idx_active_cycles=round(o['SQ_LDS_IDX_ACTIVE']))
        if o.get('SQ_LDS_IDX_ACTIVE'):
            e['lds_conflict_frac'] = round(o['SQ_LDS_BANK_CONFLICT'] / o['SQ_LDS_IDX_ACTIVE'], 4)
    if 'GRBM_GUI_ACTIVE' in o:
        e['gui_active_cycles_all_xcds'] = round(o['GRBM_GUI_ACTIVE'])
    if dur.get(k) and 'SQ_VALU_MFMA_BUSY_CYCLES' in o:
        us = sum(dur[k].values()) / len(dur[k]) / 1e3
        e['us_per_launch_profiled'] = round(us, 2)
        # MFMA-busy cycles are summed over the 1024 SIMDs of the chip: utilisation = busy / (1024 x elapsed cycles); the
        # elapsed cycles are taken at the 2.4-GHz maximum clock (profiled passes clock lower: a lower bound)
        e['mfma_pipe_util'] = round(o['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * us * 2400.0), 4)
    if 'FETCH_SIZE' in o:
        e['hbm_read_bytes_per_launch'] = int(2 * o['FETCH_SIZE'] * 1024)
    if 'WRITE_SIZE' in o:
        e['hbm_write_bytes_per_launch'] = int(o['WRITE_SIZE'] * 1024)
    if 'FETCH_SIZE' in o and 'WRITE_SIZE' in o:
        e['hbm_bytes_per_launch'] = e['hbm_read_bytes_per_launch'] + e['hbm_write_bytes_per_launch']
    if 'TCC_HIT_sum' in o:
        e['l2_hit_rate'] = round(o['TCC_HIT_sum'] / max(o['TCC_HIT_sum'] + o['TCC_MISS_sum'], 1.0), 4)
    out[k] = e
note = ('rocprofv3 --kernel-trace --pmc <one counter set per pass> over scripts/roofline_kernels.py (scripts/pmc_round.sh): per-launch averages, '
        'kernels launched in isolation on one stream; SQ_* in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs); FETCH_SIZE x2.')
json.dump(dict(_note=note, **out), open('profiles/%s_pmc_kernels.json' % tag, 'w'), indent=1)
json.dump(dict(_note=note, **{k: v for k, v in out.items() if k in ('tn128g', 'tn128', 'tn256', 'nt128', 'nt256', 'k_splitk_reduce', 'k_splitk_reduce_group')}), open('profiles/%s_pmc_gemm.json' % tag, 'w'), indent=1)
for k, v in out.items():
    print(k, v)
