"""profiles/r01b_pmc_gemm_gx.json from the two rocprofv3 --pmc passes over scripts/roofline_gemm.py
(FETCH_SIZE and WRITE_SIZE collected separately; FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md HBM section)."""
import csv, sys, json, glob
def avg(path, counter, shape_grid):
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(path))
         if 'k_gemm_nt' in r['Kernel_Name'] and r['Counter_Name'] == counter and int(r['Grid_Size']) == shape_grid]
    return sum(v) / len(v), len(v)
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 442 * 512            # workgroups x threads of the instance's launch
f, n1 = avg(sys.argv[1], 'FETCH_SIZE', grid)
w, n2 = avg(sys.argv[2], 'WRITE_SIZE', grid)
out = {'_note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over scripts/roofline_gemm.py; per-launch averages of the '
                'M=8704 N=3200 K=800 instance (256x256 tiles: grid 442 x 512 threads); counter unit KB; FETCH_SIZE doubled per MI355X_MICROARCH.md',
       'k_gemm_nt': {'FETCH_SIZE_KB': round(f, 1), 'WRITE_SIZE_KB': round(w, 1), 'hbm_read_bytes': int(2 * f * 1024),
                     'hbm_write_bytes': int(w * 1024), 'launches_sampled': [n1, n2]}}
print(json.dumps(out, indent=1))
