"""From a rocprofv3 kernel trace of bench.py: how much of the wall time has NO kernel running (union over all streams),
and where the gaps are.  usage: idle_gaps.py trace.csv [from_fraction to_fraction]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t_first, t_last = rows[0][0], max(r[1] for r in rows)
# the timed steps are the longest stretch of the trace without a gap of 4 ms or more (set-up, video synthesis, the untimed
# extra passes and the event read-back all leave longer holes)
segs, start, cur_end = [], 0, rows[0][1]
for i, (st, en, _) in enumerate(rows):
    if st - cur_end >= 4e6:
        segs.append((start, i)); start = i
    cur_end = max(cur_end, en)
segs.append((start, len(rows)))
a, b = max(segs, key=lambda ab: rows[ab[1] - 1][1] - rows[ab[0]][0])
sel = rows[a:b]
busy, gaps, cur_end, prev_name = 0, [], sel[0][0], ""
for s, e, n in sel:
    if s > cur_end:
        gaps.append((s - cur_end, prev_name[:50], n[:50]))
        busy += 0
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
        prev_name = n
span = cur_end - sel[0][0]
print("span %.1f ms, busy (union) %.1f ms = %.1f %%, idle %.1f ms in %d gaps" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps)))
big = sorted(gaps, reverse=True)[:12]
for g, a, b in big:
    print("  gap %.3f ms after [%s] before [%s]" % (g / 1e6, a, b))
import collections
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else "<1ms" if g < 1e6 else ">=1ms"] += g
print({k: round(v / 1e6, 2) for k, v in hist.items()})
