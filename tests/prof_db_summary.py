"""Summarise a rocprofv3 --kernel-trace results .db (rocpd sqlite schema): per-kernel totals, like --stats' CSV.
    python tests/prof_db_summary.py <results.db> [divide_by_steps] [rows]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                        f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# {sys.argv[1]}\n# total kernel time {tot/1e6:.2f} ms over the run; per-step column divides by {div:g}")
print(f"{'total_ms':>10} {'per_step_ms':>11} {'pct':>6} {'calls':>6} {'avg_us':>9}  kernel")
for name, n, t, mn, mx in rows[:top]:
    short = re.sub(r"\(.*", "", name)[:110]
    print(f"{t/1e6:10.2f} {t/1e6/div:11.2f} {100*t/tot:6.2f} {n:6d} {t/n/1e3:9.1f}  {short}")
