"""Per-kernel means of a rocprofv3 --pmc CSV directory (counter_collection.csv)."""
import collections
import csv
import glob
import os
import re
import sys


def summarize(d, only="kpl"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    files = sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:          # gpurun_out/ accumulates the files of earlier runs: newest only
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if only and only not in name:
                continue
            m = re.search(r"(\w+(?:<[^>]*>)?)\((?!anonymous)", name)
            short = m.group(1) if m else name[:40]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[short] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"],
                           r["Scratch_Size"], r["Workgroup_Size"], r["Grid_Size"])
    return agg, meta


if __name__ == "__main__":
    agg, meta = summarize(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "kpl")
    for k in sorted(agg):
        print(k, "vgpr/agpr/sgpr/lds/scratch/wg/grid =", meta[k])
        for c in sorted(agg[k]):
            v = agg[k][c]
            print("    %-28s n=%-3d mean=%.1f" % (c, len(v), sum(v) / len(v)))
