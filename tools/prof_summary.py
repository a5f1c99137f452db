"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as a per-kernel table:
   python tools/prof_summary.py gpurun_out/prof_r01/r01_results.db > profiles/r01_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                        "from kernels group by name order by sum(end-start) desc"))
total = sum(r[2] for r in rows) or 1
print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for name, n, tot, avg, mn, mx in rows:
    short = name if len(name) <= 72 else name[:69] + "..."
    print(f"{short:72s} {n:7d} {tot/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*tot/total:6.2f}")
print(f"{'TOTAL':72s} {sum(r[1] for r in rows):7d} {total/1e6:10.3f}")
