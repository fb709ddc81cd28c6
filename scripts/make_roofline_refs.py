"""profiles/roofline_refs.json: what bench.py's `roofline` quotes beside its live (isolated) timing --
the IN-STEP average duration of each GEMM kernel row of a committed rocprofv3 `--kernel-trace --stats` summary of the train step,
and the HBM bytes per launch from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md).
    python scripts/make_roofline_refs.py <cfg> <profiles/<tag>_kernel_stats.csv> [<profiles/<tag>_pmc_gemm.json>]
Entries are merged into the existing file per config."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg, stats = sys.argv[1], sys.argv[2]
pmc = sys.argv[3] if len(sys.argv) > 3 else None
ROWS = {'tn128g': 'k_gemm_tn_group', 'tn128': 'k_gemm_nt<128, 128, 2, 2, true, true', 'nt128': 'k_gemm_nt<128, 128, 2, 2, true, false',
        'nt256': 'k_gemm_nt<256, 256, 2, 4, false, false', 'tn256': 'k_gemm_nt<256, 256, 2, 4, false, true'}
out_path = os.path.join(ROOT, 'profiles', 'roofline_refs.json')
refs = json.load(open(out_path)) if os.path.exists(out_path) else {}
ent = {}
rows = list(csv.DictReader(open(stats)))
for inst, pat in ROWS.items():
    for r in rows:
        if pat in r['Name']:
            ent[inst] = dict(in_step_us_per_launch=round(float(r['AverageNs']) / 1e3, 2), in_step_calls=int(r['Calls']),
                             in_step_source=os.path.relpath(stats, ROOT))
            break
if pmc:
    pj = json.load(open(pmc))
    for inst in ent:
        if inst in pj and 'hbm_bytes_per_launch' in pj[inst]:
            ent[inst]['hbm_bytes_per_launch'] = pj[inst]['hbm_bytes_per_launch']
            ent[inst]['traffic_source'] = os.path.relpath(pmc, ROOT)
refs[cfg] = ent
json.dump(refs, open(out_path, 'w'), indent=1, sort_keys=True)
print(json.dumps(ent, indent=1))
