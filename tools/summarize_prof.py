#!/usr/bin/env python
"""Condenses rocprofv3 output directories into the small text summaries committed under profiles/.

    python tools/summarize_prof.py stats <dir> > profiles/rNN_<what>_kernel_stats.txt
    python tools/summarize_prof.py pmc   <dir> > profiles/rNN_<what>_pmc.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))


def stats(d):
    files = find(d, "*kernel_stats.csv")
    if not files:
        print("no *kernel_stats.csv under", d)
        return
    for f in files:
        print("#", os.path.relpath(f, d))
        rows = list(csv.DictReader(open(f)))
        print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>11s} {'avg_us':>11s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for r in rows[:40]:
            name = r.get("Name", "")[:70]
            print(f"{name:70s} {r.get('Calls',''):>7s} {float(r.get('TotalDurationNs',0))/1e6:11.3f} "
                  f"{float(r.get('AverageNs',0))/1e3:11.2f} {float(r.get('MinNs',0))/1e3:10.2f} "
                  f"{float(r.get('MaxNs',0))/1e3:10.2f} {r.get('Percentage',''):>6s}")


def pmc(d):
    files = find(d, "*counter_collection.csv")
    if not files:
        print("no *counter_collection.csv under", d)
        return
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"{'kernel':70s} {'counter':>14s} {'launches':>9s} {'mean':>16s} {'min':>16s} {'max':>16s}")
    for k in sorted(acc):
        for c, v in sorted(acc[k].items()):
            print(f"{k:70s} {c:>14s} {len(v):9d} {sum(v)/len(v):16.1f} {min(v):16.1f} {max(v):16.1f}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
