"""rocprofv3 --kernel-trace database (rocpd .db) -> per-kernel summary CSV (the table `--stats` prints).

    python tools/kernel_stats_from_db.py gpurun_out/prof/bench_results.db profiles/r01_kernel_stats.csv
"""
import csv, sqlite3, sys

db, out = sys.argv[1:3]
con = sqlite3.connect(db)
rows = list(con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                        "from kernels group by name order by 3 desc"))
total = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name, calls, tot, round(avg, 1), round(100.0 * tot / total, 3), mn, mx])
print(f"{len(rows)} kernels, {total / 1e6:.3f} ms of kernel time -> {out}")
