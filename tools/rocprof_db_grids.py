"""Kernel durations of a rocprofv3 rocpd database grouped by (kernel, grid): which launch shapes a family's time sits in.
python tools/rocprof_db_grids.py x_results.db [name-substring]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm_nt"
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
gx = "grid_size_x" if "grid_size_x" in cols else "grid_x"
wx = "workgroup_size_x" if "workgroup_size_x" in cols else "workgroup_x"
gy = gx.replace("x", "y")
rows = c.execute(f"select name, {gx}, {gy}, {wx}, count(*), avg(end - start), sum(end - start), min(end - start) from kernels where name like ? group by name, {gx}, {gy} order by sum(end - start) desc", ("%" + pat + "%",)).fetchall()
tot = sum(r[6] for r in rows)
print("%d launch shapes, %.1f ms" % (len(rows), tot / 1e6))
for name, x, y, w, n, avg, s, mn in rows[:60]:
    short = name[name.find("<"):name.find(">") + 1] if "<" in name else name[:40]
    print("%5.1f %%  %5d x %8.1f us (min %7.1f)  blocks %6d x %d  %s" % (100.0 * s / tot, n, avg / 1e3, mn / 1e3, x // max(w, 1), y, short))
