"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv (utility; used to produce profiles/*.txt)."""
import csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"# {files[0]}\n# total kernel time {tot/1e6:.2f} ms over the run; per-step column divides by {div:g}")
print(f"{'total_ms':>10} {'per_step_ms':>11} {'pct':>6} {'calls':>6} {'avg_us':>9}  kernel")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{float(r['TotalDurationNs'])/1e6:10.2f} {float(r['TotalDurationNs'])/1e6/div:11.2f} {float(r['Percentage']):6.2f} {r['Calls']:>6} {float(r['AverageNs'])/1e3:9.1f}  {r['Name'][:110]}")
