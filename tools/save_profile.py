"""Copies the judged summaries of a profiling run from gpurun_out/ (scratch) into profiles/ (tracked):
   python tools/save_profile.py <tag> <trace_dir> <pmc_fetch_dir> <pmc_write_dir> [bench_json]
Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and refreshes profiles/traffic.json
(what bench.py reports as roofline.traffic): HBM bytes per launch of the score kernel =
2 x FETCH_SIZE (gfx950 counts a 16-B-per-lane read at half its bytes, MI355X_MICROARCH.md "HBM")
+ WRITE_SIZE, both in KiB from separate --pmc passes."""
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarize  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, trace, fetch, write = sys.argv[1:5]
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(trace, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(prof, tag + "_kernel_stats.csv"))
    out = {}
    for name, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        agg, meta = summarize(d)
        for k, v in agg.items():
            vals = v.get(name, [])
            if vals:
                out.setdefault(k, {})[name + "_KiB_mean"] = sum(vals) / len(vals)
                out[k]["launches_" + name] = len(vals)
                out[k]["vgpr/agpr/sgpr/lds/scratch/wg/grid"] = meta[k]
    json.dump(out, open(os.path.join(prof, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    sk = out.get("score_kernel<false>", {})
    views = 1
    if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
        try:
            views = int(json.load(open(sys.argv[5]))["config"].get("views_per_launch", 1))
        except Exception:
            pass
    if "FETCH_SIZE_KiB_mean" in sk and "WRITE_SIZE_KiB_mean" in sk:
        traffic = (2.0 * sk["FETCH_SIZE_KiB_mean"] + sk["WRITE_SIZE_KiB_mean"]) * 1024.0
        json.dump({"score_kernel_hbm_bytes_per_launch": int(traffic), "views_per_launch": views, "source": tag + "_pmc.json",
                   "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate rocprofv3 --pmc passes"},
                  open(os.path.join(prof, "traffic.json"), "w"), indent=1)
    if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
        shutil.copy(sys.argv[5], os.path.join(prof, tag + "_bench.json"))
    print("saved", tag, "->", prof)


if __name__ == "__main__":
    main()
