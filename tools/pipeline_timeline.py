#!/usr/bin/env python3
"""One steady-state period of the streamed pipeline from a rocprofv3 trace: tools/pipeline_timeline.py <prefix>
(<prefix>_kernel_trace.csv and, when there, <prefix>_memory_copy_trace.csv of `rocprofv3 --kernel-trace --memory-copy-trace --
python tools/pipeline_bench.py --depth 4 ...`).  Prints what ran between two decodes in the middle of the run (us from the first
decode's start, duration, queue), the busy time of every queue over the period and the period itself."""
import csv
import os
import sys

pre = sys.argv[1]
ev = []
for r in csv.DictReader(open(pre + "_kernel_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q%s" % r.get("Queue_Id", "?"), r["Kernel_Name"].replace("void ", "").split("(")[0]))
mc = pre + "_memory_copy_trace.csv"
if os.path.exists(mc):
    for r in csv.DictReader(open(mc)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", "%s %.1f MB" % (r.get("Direction", "copy"), int(r.get("Size", 0) or 0) / 1e6)))
ev.sort()
dec = [e for e in ev if e[3].startswith("jda_decode_tiles")]
if len(dec) < 6:
    sys.exit("too few decodes in the trace")
a, b = dec[len(dec) // 2], dec[len(dec) // 2 + 1]
t0, t1 = a[0], b[0]
print("period %.1f us (decode to decode), decode %.1f us" % ((t1 - t0) / 1e3, (a[1] - a[0]) / 1e3))
busy = {}
for s, e, q, n in ev:
    if e <= t0 or s >= t1:
        continue
    busy[q] = busy.get(q, 0) + (min(e, t1) - max(s, t0))
    if (e - s) > 3000:
        print("  %9.1f us + %8.1f  %-5s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:60]))
print("busy per queue over the period: " + ", ".join("%s %.0f us" % (q, v / 1e3) for q, v in sorted(busy.items())))
periods = [(dec[i + 1][0] - dec[i][0]) / 1e3 for i in range(len(dec) // 4, len(dec) - 2)]
print("periods of the steady state: mean %.1f us, min %.1f, max %.1f (%d)" % (sum(periods) / len(periods), min(periods), max(periods), len(periods)))
