import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in files:
    for r in csv.DictReader(open(fn)):
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'gemm' not in k and 'attn' not in k and 'ln_' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
