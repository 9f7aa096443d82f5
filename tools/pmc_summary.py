"""Per-kernel average of rocprofv3 --pmc counters: python tools/pmc_summary.py gpurun_out/pmc_*"""
import csv, glob, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); name = re.sub(r'\(.*', '', name).replace('void ', '')
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
keys = sorted({k for v in agg.values() for k in v})
print('kernel | launches | ' + ' | '.join(keys))
rows = []
for name, cs in agg.items():
    n = max(len(v) for v in cs.values())
    rows.append((name, n, [sum(cs[k]) / len(cs[k]) if k in cs else float('nan') for k in keys]))
rows.sort(key=lambda r: -r[1] * (r[2][keys.index('FETCH_SIZE')] if 'FETCH_SIZE' in keys else 1))
for name, n, vals in rows[:int(sys.argv[0] and 24)]:
    print(f"{name[:44]:44s} | {n:5d} | " + ' | '.join(f"{v:12.4g}" for v in vals))
