#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a small table for profiles/.
usage: summarize_profile.py <kernel_stats.csv> <steps incl. warm-up> <out.md> [title]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
title = sys.argv[4] if len(sys.argv) > 4 else "rocprofv3 --kernel-trace --stats"
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(sys.argv[3], "w") as f:
    f.write(f"# {title}\n\nsource: `{sys.argv[1].split('gpurun_out/')[-1]}`, {int(steps)} steps (warm-up included in the averages)\n\n")
    f.write("| kernel | calls/step | avg us/call | ms/step | % of kernel time |\n|---|---|---|---|---|\n")
    for r in rows[:40]:
        t = float(r["TotalDurationNs"])
        f.write(f"| `{r['Name'][:110]}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {t / steps / 1e6:.3f} | {100 * t / tot:.2f} |\n")
    f.write(f"\ntotal kernel time per step: {tot / steps / 1e6:.2f} ms\n")
