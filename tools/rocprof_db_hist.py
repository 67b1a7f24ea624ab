"""Duration histogram of the launches of one kernel (name substring) in a rocprofv3 rocpd database, with the grid of the launches in each
bucket: what a family of launches consists of (e.g. `__amd_rocclr_copyBuffer`: tiny D2H / H2D blits vs the ring carries).
python tools/rocprof_db_hist.py x_results.db copyBuffer"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
gx = "grid_size_x" if "grid_size_x" in cols else "grid_x"
rows = c.execute(f"select end - start, {gx} from kernels where name like ?", ("%" + pat + "%",)).fetchall()
tot_all = c.execute("select sum(end - start) from kernels").fetchone()[0]
edges = [2e3, 5e3, 1e4, 2e4, 5e4, 1e5, 2e5, 5e5, 1e6, 1e12]
hist = collections.OrderedDict((e, [0, 0.0, collections.Counter()]) for e in edges)
for d, g in rows:
    for e in edges:
        if d < e:
            h = hist[e]
            h[0] += 1; h[1] += d; h[2][g] += 1
            break
tot = sum(d for d, _ in rows)
print("%s: %d launches, %.2f ms = %.2f %% of all kernel time" % (pat, len(rows), tot / 1e6, 100.0 * tot / max(tot_all, 1)))
lo = 0
for e, (n, t, grids) in hist.items():
    if n:
        print("  %7.0f - %7.0f us: %5d launches, %8.2f ms (%5.1f %%)  grids %s" % (lo / 1e3, e / 1e3 if e < 1e12 else float("inf"), n, t / 1e6, 100.0 * t / max(tot, 1),
                                                                             dict(grids.most_common(4))))
    lo = e
