"""Timeline of ONE train step from a rocprofv3 --kernel-trace CSV (kernels between two consecutive k_adam_ema)."""
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
tr.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with k_inc_step + the final optimiser launch(es) (earlier k_adam_ema launches of a step belong to ranges that
# are updated under the remaining backward stages)
ends = []
for i, r in enumerate(tr):
    if r['Kernel_Name'].startswith('k_inc_step'):
        j = i
        while j + 1 < len(tr) and tr[j + 1]['Kernel_Name'].startswith('k_adam_ema'):
            j += 1
        ends.append(j)
seg = tr[ends[-5] + 1:ends[-4] + 1]
t0 = int(seg[0]['Start_Timestamp'])
busy = 0.0
for r in seg:
    nm = r['Kernel_Name'].split('(')[0][:34]
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    busy += e - s
    if e - s > thr or 'persist' in nm:
        print('%8.1f %8.1f  %6.1f  q%s %-36s grid %5d x %s' % (s, e, e - s, r['Queue_Id'], nm, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y']))
print('kernels %d, summed kernel time %.1f us' % (len(seg), busy))
