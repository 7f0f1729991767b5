"""What runs next to what: from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) of `bench.py --lean` the share of the
wall time with 0 / 1 / 2+ kernels in flight, the time every kernel spends ALONE on the GPU, and per stream the average
gap between the end of a kernel and the start of the next one.
    python tools/timeline.py <kernel_trace.csv> [skip-first-fraction]"""
import collections
import csv
import sys


def short(n):
    n = n.replace("void ", "").split("(anonymous namespace)::")[-1]
    return n.split("(")[0]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows))
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * skip          # leave the warm-up and the extras at the start out
    ev = [e for e in ev if e[0] >= t_lo]
    points = []
    for s, e, k, q in ev:
        points.append((s, 1, k))
        points.append((e, -1, k))
    points.sort()
    level, last = 0, points[0][0]
    share = collections.Counter()
    running = collections.Counter()
    alone = collections.Counter()
    for t, d, k in points:
        dt = t - last
        share[min(level, 2)] += dt
        if level == 1:
            alone[next(iter(+running))] += dt
        level += d
        running[k] += d
        last = t
    wall = points[-1][0] - points[0][0]
    print("wall %.2f ms; kernels in flight: none %.1f %%, one %.1f %%, two or more %.1f %%"
          % (wall / 1e6, 100 * share[0] / wall, 100 * share[1] / wall, 100 * share[2] / wall))
    dur = collections.Counter()
    for s, e, k, q in ev:
        dur[k] += e - s
    print("%-40s %10s %10s" % ("kernel", "busy % wall", "alone % wall"))
    for k, v in dur.most_common(12):
        print("%-40s %10.1f %10.1f" % (k[:40], 100 * v / wall, 100 * alone[k] / wall))
    by_q = collections.defaultdict(list)
    for s, e, k, q in ev:
        by_q[q].append((s, e, k))
    for q, lst in by_q.items():
        gaps = [b[0] - a[1] for a, b in zip(lst, lst[1:])]
        print("queue %s: %d kernels, mean gap %.1f us, gaps > 20 us: %d" % (q, len(lst), sum(gaps) / max(len(gaps), 1) / 1e3, sum(g > 20000 for g in gaps)))


if __name__ == "__main__":
    main()
