"""rocprofv3 (ROCm 7.2) writes a rocpd sqlite database by default: turn its `top_kernels` view into the kernel-stats CSV kept
under profiles/ and print the shares the round's targets are stated in (torch `at::native` / hipBLASLt `Cijk` kernels by time
and by launches).   python tools/rocprof_db_stats.py gpurun_out/prof/x_results.db profiles/rNN_bench_kernel_stats.csv"""
import csv
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
    tot_t = sum(r[2] for r in rows)
    tot_n = sum(r[1] for r in rows)
    if out:
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
            for r in rows:
                w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.4f" % (100.0 * r[2] / tot_t)])
    fam = {"at::native + other torch": lambda n: "at::native" in n or n.startswith("void at::") or "elementwise" in n,
           "hipBLASLt Cijk": lambda n: n.startswith("Cijk"),
           "ours: gemm": lambda n: "gemm_nt" in n, "ours: msda": lambda n: "msda" in n}
    print("total: %d launches, %.1f ms of kernel time" % (tot_n, tot_t / 1e3))      # (the view reports microseconds)
    for k, fn in fam.items():
        t = sum(r[2] for r in rows if fn(r[0]))
        n = sum(r[1] for r in rows if fn(r[0]))
        print("%-26s %6.2f %% of time  %6.2f %% of launches (%d)" % (k, 100.0 * t / tot_t, 100.0 * n / tot_n, n))
    for r in rows[:25]:
        print("%6.2f %%  %6d x %9.1f us  %s" % (100.0 * r[2] / tot_t, r[1], r[3], r[0][:110]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
