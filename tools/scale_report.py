"""tools/scale_report.py <bench line files...>: reads the JSON lines `bench.py --gpus N` printed for N = 1, 2, 4, 8 (one file per
N, or one file with several lines; the driver's SCALE_rNN.json works too if it holds them under "runs") and prints, per N:
whole-job throughput, efficiency against N = 1 (weak scaling: value_N / (N x value_1)), the spread of the ranks' own times
(per_rank_ms_per_step: max / min), and the exchange step's share (collective.ms_per_step / ms_per_step -- the gather runs on a
stream of its own beside the other batch's scoring, so a share below 1 costs nothing).

No scaling curve has been measured on hardware yet (one GPU per gpurun call): this only makes the day a node exists a
one-command day.  `python tools/scale_report.py --self-test` checks the arithmetic on made-up lines."""
import json
import sys


def lines_of(path):
    rows = []
    text = open(path).read().strip()
    try:
        doc = json.loads(text)
        if isinstance(doc, dict) and "runs" in doc:
            doc = doc["runs"]
        if isinstance(doc, dict) and "metric" in doc:
            doc = [doc]
        if isinstance(doc, list):
            for r in doc:
                r = r.get("parsed", r) if isinstance(r, dict) else r
                if isinstance(r, dict) and "n_gpus" in r:
                    rows.append(r)
            return rows
    except ValueError:
        pass
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{"):
            try:
                r = json.loads(ln)
            except ValueError:
                continue
            if "n_gpus" in r:
                rows.append(r)
    return rows


def report(rows):
    by_n = {}
    for r in rows:
        by_n[int(r["n_gpus"])] = r          # (the last line for an N wins)
    if 1 not in by_n:
        raise SystemExit("no N = 1 line: efficiency needs it")
    base = by_n[1]["value"]
    out = []
    for n in sorted(by_n):
        r = by_n[n]
        per = r.get("per_rank_ms_per_step") or []
        coll = (r.get("collective") or {}).get("ms_per_step")
        out.append({"n_gpus": n, "value": r["value"], "unit": r.get("unit"), "ms_per_step": r["ms_per_step"],
                    "efficiency_vs_1": round(r["value"] / (n * base), 4),
                    "rank_spread_max_over_min": round(max(per) / min(per), 4) if per else None,
                    "slowest_rank": int(max(range(len(per)), key=lambda k: per[k])) if per else None,
                    "collective_ms_per_step": coll,
                    "collective_share_of_step": round(coll / r["ms_per_step"], 4) if coll else None,
                    "devices_shared": r.get("devices_shared")})
    return out


def self_test():
    mk = lambda n, v, ms, per=None, c=None: {"metric": "m", "n_gpus": n, "value": v, "unit": "Mpoints/s", "ms_per_step": ms,
                                            "per_rank_ms_per_step": per, "collective": {"ms_per_step": c} if c else None}
    rows = report([mk(1, 2000.0, 0.8), mk(2, 3900.0, 0.82, [0.82, 0.80], 0.05), mk(8, 15200.0, 0.842, [0.84, 0.8, 0.81, 0.82, 0.83, 0.8, 0.8, 0.842], 0.09)])
    assert [r["n_gpus"] for r in rows] == [1, 2, 8]
    assert rows[1]["efficiency_vs_1"] == 0.975 and rows[2]["efficiency_vs_1"] == 0.95
    assert rows[1]["rank_spread_max_over_min"] == 1.025 and rows[2]["slowest_rank"] == 7
    assert rows[2]["collective_share_of_step"] == round(0.09 / 0.842, 4)
    print("scale_report self-test ok")
    return 0


if __name__ == "__main__":
    if "--self-test" in sys.argv:
        sys.exit(self_test())
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    if not files:
        print(__doc__)
        sys.exit(2)
    rows = []
    for f in files:
        rows += lines_of(f)
    rep = report(rows)
    if "--json" in sys.argv:
        print(json.dumps(rep))
    else:
        print("%6s %14s %12s %11s %12s %16s %10s" % ("N", "value", "ms/step", "efficiency", "rank spread", "collective ms", "share"))
        for r in rep:
            print("%6d %14.1f %12.5f %11.4f %12s %16s %10s" % (r["n_gpus"], r["value"], r["ms_per_step"], r["efficiency_vs_1"],
                                                            r["rank_spread_max_over_min"], r["collective_ms_per_step"],
                                                            r["collective_share_of_step"]))
