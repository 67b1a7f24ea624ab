"""Whole-GPU idle gaps of the last bench step in a rocprofv3 --kernel-trace --hip-trace database, with the host HIP calls that
overlap each gap.   python tools/trace_gaps.py <results.db> [min_gap_us]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 200e3
ks = c.execute("select start, end, name from kernels order by start").fetchall()
try:
    mc = c.execute("select start, end from memory_copies order by start").fetchall()
except Exception:
    mc = []
stems = [k[0] for k in ks if "stem_conv" in k[2]]
a = stems[-4] if len(stems) >= 4 else ks[0][0]
b = max(k[1] for k in ks)
iv = sorted([(k[0], k[1]) for k in ks if k[1] > a] + [(m[0], m[1]) for m in mc if m[1] > a])
busy, ce, gaps = 0, a, []
for s, e in iv:
    if s > ce:
        if s - ce > min_gap:
            gaps.append((ce, s))
    busy += max(0, e - max(s, ce))
    ce = max(ce, e)
print("last step: wall %.1f ms, GPU busy (kernels + copies, union) %.1f ms = %.1f %%" % ((b - a) / 1e6, busy / 1e6, 100.0 * busy / (b - a)))
cols = [r[1] for r in c.execute("pragma table_info('regions')")]
print("gaps > %.0f us: %d, total %.1f ms" % (min_gap / 1e3, len(gaps), sum(e - s for s, e in gaps) / 1e6))
for s, e in gaps[:40]:
    rows = c.execute("select name, start, end from regions where start < ? and end > ? order by (min(end, ?) - max(start, ?)) desc limit 4", (e, s, e, s)).fetchall()
    desc = "; ".join("%s %.0fus" % (r[0], (r[2] - r[1]) / 1e3) for r in rows)
    print("  t=%.1f ms  gap %.0f us : %s" % ((s - a) / 1e6, (e - s) / 1e3, desc))
