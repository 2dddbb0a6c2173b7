"""Idle time of the stream between the kernels of a step, from a rocprofv3 --kernel-trace CSV:
   python tools/trace_gaps.py <kernel_trace.csv> [first kernel name fragment of a step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "gram_direct"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
starts = [i for i, e in enumerate(ev) if first in e[2]]
starts = starts[len(starts) // 2:]                       # the timed region's steps
per = {}
nstep = 0
for a, b in zip(starts[:-1], starts[1:]):
    nstep += 1
    for i in range(a, b):
        nm = ev[i][2].split("(")[0][-40:]
        busy = ev[i][1] - ev[i][0]
        gap = ev[i + 1][0] - ev[i][1]
        p = per.setdefault((i - a, nm), [0, 0]); p[0] += busy; p[1] += gap
tb = tg = 0.0
for (k, nm), (busy, gap) in sorted(per.items()):
    print("%2d %-42s busy %7.2f us   gap after %6.2f us" % (k, nm, busy / nstep / 1e3, gap / nstep / 1e3))
    tb += busy / nstep / 1e3; tg += gap / nstep / 1e3
print("per step: busy %.2f us, idle %.2f us, total %.2f us (%d steps)" % (tb, tg, tb + tg, nstep))
