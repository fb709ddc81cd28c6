"""Average per-launch PMC counter values per kernel from a rocprofv3 --pmc CSV."""
import csv, collections, sys, json
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, v in sorted(agg.items()):
    out[k] = {c: sum(x) / len(x) for c, x in v.items()}
    out[k]['launches'] = len(next(iter(v.values())))
print(json.dumps(out, indent=1))
