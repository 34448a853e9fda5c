#!/usr/bin/env python
"""Timeline of ONE hipGraph replay from a rocprofv3 --kernel-trace csv: how much of the step is the
device busy, how much is idle between dependent kernels, how much runs concurrently.
usage: graph_timeline.py KERNEL_TRACE.csv KERNELS_PER_STEP_HINT"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r.get("Start_Timestamp")]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# steps: split at adam_multi_kernel occurrences is fragile; use the LAST `n` kernels where n = hint
n = int(sys.argv[2])
# bench.py runs its calibration kernels after the timed region: the step is the last `n` kernels
# in front of the first calibration launch
cal = [i for i, r in enumerate(rows) if "calib_" in r["Kernel_Name"]]
if cal:
    rows = rows[:cal[0]]
    while rows and not rows[-1]["Kernel_Name"].endswith("counter_add_kernel(long*, long)"):
        rows.pop()      # (fills / copies between the last replay and the calibration)
step = rows[-n:]
t0, t1 = step[0]["s"], max(r["e"] for r in step)
span = (t1 - t0) / 1e3
busy_sum = sum(r["e"] - r["s"] for r in step) / 1e3
# union of busy intervals
iv = sorted((r["s"], r["e"]) for r in step)
union, cs, ce = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > ce:
        union += ce - cs
        gaps.append((s - ce, ce))
        cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
print("kernels %d  span %.1f us  sum of durations %.1f us  union busy %.1f us  idle %.1f us (%d gaps, median %.2f us)" % (
    len(step), span, busy_sum, union / 1e3, span - union / 1e3, len(gaps),
    sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0))
hist = defaultdict(int)
for g, _ in gaps:
    hist[min(int(g / 1e3), 20)] += 1
print("gap histogram (us: count):", dict(sorted(hist.items())))
by = defaultdict(lambda: [0, 0.0])
for r in step:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    by[k][0] += 1
    by[k][1] += (r["e"] - r["s"]) / 1e3
print("top kernels of the step:")
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-62s %4d %9.1f us" % (k, c, t))
# largest gaps with the kernels on either side
idx = {r["e"]: r for r in step}
print("largest gaps:")
for g, ce_ in sorted(gaps, reverse=True)[:12]:
    prev = idx.get(ce_)
    nxt = next((r for r in step if r["s"] >= ce_ + g), None)
    print("  %.1f us after %-40s before %s" % (g / 1e3, (prev or {}).get("Kernel_Name", "?")[-40:],
                                               (nxt or {}).get("Kernel_Name", "?")[-50:]))
