"""Summarise a rocprofv3 kernel_stats.csv: python tools/prof_summary.py <dir> <steps> [out.md]"""
import csv, glob, re, sys
d, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
lines = [f"total kernel time {tot/1e6:.1f} ms over {steps} steps = {tot/steps/1e6:.2f} ms/step", "",
         "| kernel | calls/step | ms/step | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:32]:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Name']); name = re.sub(r'\(.*', '', name).replace('void ', '')
    lines.append(f"| `{name}` | {int(r['Calls'])/steps:.1f} | {float(r['TotalDurationNs'])/steps/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], 'a').write(out + "\n")
