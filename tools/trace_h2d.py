"""Where the host->device copies of a video sit relative to its kernels (rocprofv3 --kernel-trace --memory-copy-trace database):
per video of the run: first copy start, when 20 / 60 / 120 frames' worth of bytes have landed, first stem kernel start, last kernel end.
python tools/trace_h2d.py <results.db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
ks = c.execute("select start, end, name from kernels order by start").fetchall()
cols = [r[1] for r in c.execute("pragma table_info('memory_copies')")]
print("memory_copies columns:", cols)
size_col = "size" if "size" in cols else ("bytes" if "bytes" in cols else None)
name_col = "name" if "name" in cols else None
mc = c.execute("select start, end, %s, %s from memory_copies order by start" % (size_col or "0", name_col or "''")).fetchall()
big = [m for m in mc if m[2] and m[2] > 3e6]          # the 10-frame upload chunks (6.9 MB)
print("%d copies, %d of them > 3 MB; kinds: %s" % (len(mc), len(big), sorted(set(m[3] for m in mc))[:6]))
stems = [k for k in ks if "stem_conv" in k[2]]
# a video's upload = a run of big copies less than 20 ms apart
runs, cur = [], [big[0]]
for m in big[1:]:
    if m[0] - cur[-1][1] > 20e6:
        runs.append(cur); cur = [m]
    else:
        cur.append(m)
runs.append(cur)
for i, grp in enumerate(runs[-5:]):
    t0 = grp[0][0]
    st = [k for k in stems if k[0] >= t0 - 1e6][:4]
    prev_end = max([k[1] for k in ks if k[1] <= t0] or [t0])
    first_k = next((k for k in ks if k[0] >= t0), None)
    last_k = max([k[1] for k in ks if k[0] >= t0 and k[0] < t0 + 400e6 and (i + 1 >= len(runs[-5:]) or k[0] < runs[-5:][i + 1][0][0])] or [t0])
    print("video: %d copies | GPU idle before the first copy %.2f ms | copy k ends at %s ms | first kernel %.2f ms (%s) | stem kernels at %s ms | last kernel ends %.2f ms" % (
        len(grp), (t0 - prev_end) / 1e6, " ".join("%.2f" % ((m[1] - t0) / 1e6) for m in grp), (first_k[0] - t0) / 1e6, first_k[2][:24],
        " ".join("%.2f" % ((k[0] - t0) / 1e6) for k in st), (last_k - t0) / 1e6))
