"""HBM traffic per launch of a kernel family from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
separate runs, as MI355X_MICROARCH.md prescribes):

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <kernel-name regex> <out.json>

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950: FETCH_SIZE reports half of wide coalesced reads; both in KiB)
"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
                agg[name].append(float(r["Counter_Value"]))
    return agg


def kernels_digest():
    """same digest as bench.py: ties the summary to the kernel sources it was measured on"""
    import hashlib
    import os
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tpu_superresolution_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    fetch_dir, write_dir, pattern, out = sys.argv[1:5]
    steps = float(sys.argv[5]) if len(sys.argv) > 5 else None
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows, tot_b, tot_n = [], 0.0, 0
    for name in sorted(fe):
        if not re.search(pattern, name) or name not in wr:
            continue
        n = min(len(fe[name]), len(wr[name]))
        b = (2.0 * sum(fe[name]) / len(fe[name]) + sum(wr[name]) / len(wr[name])) * 1024.0
        rows.append({"kernel": name, "launches": n, "bytes_per_launch": b,
                     "read_bytes_per_launch": 2.0 * sum(fe[name]) / len(fe[name]) * 1024.0,
                     "write_bytes_per_launch": sum(wr[name]) / len(wr[name]) * 1024.0})
        tot_b += b * n
        tot_n += n
    res = {"family": pattern, "kernels_sha": kernels_digest(), "avg_hbm_bytes_per_launch": tot_b / max(tot_n, 1), "launches_sampled": tot_n,
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                     "(gfx950: FETCH_SIZE reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section)",
           "per_kernel": rows}
    if steps:      # whole-step view: every launch of every matched kernel, divided by the number of profiled steps
        total = 0.0
        for name in fe:
            if re.search(pattern, name) and name in wr:
                total += (2.0 * sum(fe[name]) + sum(wr[name]) * len(fe[name]) / max(len(wr[name]), 1)) * 1024.0
        res["hbm_bytes_per_step"] = total / steps
        res["steps_profiled"] = steps
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_kernel"}))
    for r in rows:
        print(f"  {r['kernel'][:70]:70s} {r['launches']:5d}  {r['bytes_per_launch'] / 1e6:8.1f} MB  (R {r['read_bytes_per_launch'] / 1e6:7.1f} / W {r['write_bytes_per_launch'] / 1e6:7.1f})")


if __name__ == "__main__":
    main()
