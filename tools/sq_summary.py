"""Developer tool: per-kernel SQ counter summary from two rocprofv3 --pmc passes (tools/_pmc_step.sh) + a kernel trace."""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
pre = sys.argv[2] if len(sys.argv) > 2 else "s_"


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0][:44]


acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for tag in ("pmcA", "pmcB"):
    for f in glob.glob(f"{root}/{pre}{tag}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, tag)].add(r["Dispatch_Id"])
dur = {}
for f in glob.glob(f"{root}/{pre}prof/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = float(r["AverageNs"]) / 1e3
print(f"{'kernel':44s} {'us':>6s} {'act%':>5s} {'wait%':>5s} {'stall%':>6s} {'VALU/SIMD':>9s} {'valu%':>5s} {'mfma%':>5s} {'ldsconf%':>8s} {'waves':>6s}")
rows = []
for k, d in acc.items():
    nA, nB = len(cnt[(k, "pmcA")]), len(cnt[(k, "pmcB")])
    if not nA or not nB or k not in dur:
        continue
    wc = d["SQ_WAVE_CYCLES"] / nA
    us = dur[k]
    cyc = us * 2.0e3                      # ~2.0 GHz under load
    valu_per_simd = d["SQ_INSTS_VALU"] / nA / 1024
    valu_util = d["SQ_ACTIVE_INST_VALU"] / nA * 4 / 1024 / cyc
    mfma_util = d["SQ_VALU_MFMA_BUSY_CYCLES"] / nB / 1024 / cyc
    conf = d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1)
    rows.append((us * nA, f"{k:44s} {us:6.1f} {100 * d['SQ_ACTIVE_INST_ANY'] / nA / wc:5.1f} {100 * d['SQ_WAIT_ANY'] / nA / wc:5.1f} "
                 f"{100 * d['SQ_WAIT_INST_ANY'] / nA / wc:6.1f} {valu_per_simd:9.0f} {100 * valu_util:5.1f} {100 * mfma_util:5.1f} {100 * conf:8.1f} "
                 f"{wc * 4 / cyc:6.0f}"))
for _, line in sorted(rows, reverse=True)[:26]:
    print(line)
