"""Developer tool: effective shader clock per kernel from a rocprofv3 run with `--pmc GRBM_GUI_ACTIVE --kernel-trace
--output-format csv`: GRBM_GUI_ACTIVE (GPU-busy cycles during the dispatch) / dispatch duration.
python tests/clock_summary.py <rocprof output dir>"""
import csv, glob, sys, collections
root = sys.argv[1]
dur = {}
for fn in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for fn in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        d = dur.get(r["Dispatch_Id"])
        if d and d[1] > 20000:   # kernels longer than 20 us
            agg[d[0][:70]].append((float(r["Counter_Value"]), d[1]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    cyc = sum(x[0] for x in v); ns = sum(x[1] for x in v)
    print(f"{k:72s} n={len(v):4d}  avg {ns/len(v)/1e3:8.1f} us   {cyc/ns:5.2f} GHz")
