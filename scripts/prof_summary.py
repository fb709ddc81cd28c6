"""Summarise a rocprofv3 --kernel-trace --stats CSV pair (run from the output dir)."""
import csv, collections, sys
d = sys.argv[1]
pref = sys.argv[2] if len(sys.argv) > 2 else 'b'
rows = list(csv.DictReader(open('%s/%s_kernel_stats.csv' % (d, pref))))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms %.3f' % (tot / 1e6))
for r in rows[:14]:
    print('%-52s calls %6s  total %9.3f ms  avg %9.2f us  %5.1f%%' % (r['Name'][:52], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                                       float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
tr = list(csv.DictReader(open('%s/%s_kernel_trace.csv' % (d, pref))))
g = collections.defaultdict(list)
for r in tr:
    if 'k_gemm_nt' in r['Kernel_Name']:
        g[(int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('GEMM by (tiles, splits):')
for k in sorted(g):
    v = g[k]
    print('  tiles %5d x %2d  n %3d  avg %8.1f us  min %8.1f' % (k[0], k[1], len(v), sum(v) / len(v), min(v)))
# bench.py's roofline loop: the trailing run of back-to-back launches of the 256x256 instance (graph replays of the encoder
# input projection alone) -- this is the per-launch time `roofline.achieved` is computed from; the same instance inside
# the train step runs next to side-stream kernels and is slower
big = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in tr if 'k_gemm_nt<256' in r['Kernel_Name'])
run = []
for s_, e_ in reversed(big):
    if run and run[-1][0] - e_ > 20000:
        if len(run) >= 20:
            break
        run = []
    run.append((s_, e_))
if len(run) >= 20:
    d = [(e_ - s_) / 1e3 for s_, e_ in run]
    print('roofline loop (k_gemm_nt<256,256,...>, %d back-to-back launches at the end of the trace): avg %.2f us  min %.2f  max %.2f' % (len(d), sum(d) / len(d), min(d), max(d)))
for name in ['k_lstm_step_fwd', 'k_lstm_step_bwd']:
    g = collections.defaultdict(list)
    for r in tr:
        if r['Kernel_Name'].startswith(name):
            g[(r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k in sorted(g):
        v = g[k]
        print(name, k, 'n', len(v), 'avg %.2f us min %.2f' % (sum(v) / len(v), min(v)))
# gaps: idle time between consecutive kernels in the last 30% of the trace (graph replays)
ts = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in tr)
n0 = int(len(ts) * 0.6)
busy = sum(e - s for s, e in ts[n0:])
span = ts[-1][1] - ts[n0][0]
print('tail of trace: span %.3f ms, busy %.3f ms (%.1f%%), kernels %d' % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(ts) - n0))
