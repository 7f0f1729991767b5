"""tools/small_kernel_stats.py <kernel_stats.csv>: avg / min / max of the launch-latency-bound kernels of a bench trace (the four
that queue behind the other batch's feature kernel: VERDICT r05 weak #7), plus the feature / forest kernels for reference."""
import csv
import sys

WANT = ["compact_scan_kernel", "cell_sort_store_kernel", "bucket_offsets_kernel", "bucket_total_kernel", "nms_kernel", "bucket_hist_kernel",
        "bucket_scatter_kernel", "feature_kernel", "forest_pair_kernel", "bbox_kernel", "grid_setup_kernel"]
rows = list(csv.DictReader(open(sys.argv[1])))
for w in WANT:
    for r in rows:
        if w in r["Name"]:
            print("%-28s calls %5s avg %8.1f min %7.1f max %8.1f us" % (w, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
