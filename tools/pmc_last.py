"""prints per-dispatch counter values of the kernels matching argv[2] in a rocprofv3 --pmc csv dir (largest grids last)"""
import collections, csv, glob, sys
import os
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]
if not f:
    sys.exit("no counter_collection.csv under " + sys.argv[1])
by = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        by.setdefault(r["Dispatch_Id"], {"grid": r["Grid_Size"]})[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in by.values():
    for k, v in d.items():
        if k != "grid":
            agg[d["grid"]][k].append(v)
for g in agg:
    print("grid", g, {k: (len(v), round(sum(v) / len(v), 1)) for k, v in agg[g].items()})
