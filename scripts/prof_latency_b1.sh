# rocprofv3 kernel trace of single-utterance decode calls (scripts/latency_b1.py): kernels of ONE call -> gpurun_out/prof_b1/timeline.txt
root=$PWD; out=$PWD/gpurun_out/prof_b1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d $out/p -o b --output-format csv -- python $root/scripts/latency_b1.py ${1:-cfg2} 20 > $out/run.txt 2> $out/err.txt
cd $root; f=$(find $out/p -name 'b_kernel_trace.csv' | head -1)
python - "$f" > $out/timeline.txt <<'PY'
import csv, sys
tr = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# calls are delimited by the length kernel that opens encode(); take one B = 1 eager call from the middle of the first block
starts = [i for i, r in enumerate(tr) if r['Kernel_Name'].startswith('k_seq_lengths_tail')]
i0, i1 = starts[15], starts[16]
seg = tr[i0:i1]
t0 = int(seg[0]['Start_Timestamp'])
busy = 0.0
for r in seg:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    busy += e - s
    print('%8.1f %8.1f %6.1f  %-44s grid %5d' % (s, e, e - s, r['Kernel_Name'].split('(')[0][:44], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])))
print('kernels %d, span %.1f us, summed kernel time %.1f us' % (len(seg), (int(seg[-1]['End_Timestamp']) - t0) / 1e3, busy))
PY
rm -rf $out/p; cat $out/run.txt | head -4; cat $out/timeline.txt
