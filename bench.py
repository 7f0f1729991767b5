#!/usr/bin/env python3
"""bench.py -- Mpoints/s scored (feature + forest + NMS) on a 200k-point 2.5D view, MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it with
torch.distributed.run (one rank per GPU, RCCL).  One "step" = one pass of the hot path
(index build + feature + forest + NMS + keypoint compaction, i.e. pcl::Keypoint::compute) over
one synthetic view that is already resident in HBM.  Views are independent, so ranks never
exchange data on the data path; with N > 1 every step ends with one RCCL all-gather of the
(padded) keypoint lists -- the only exchange the path has -- and scaling is weak.

Workload = BASELINE.json configs[1]: single 200k-pt synthetic 2.5D view (tools/synth.py, seed
1 + rank), 10-tree forest data/forests/synth200k_a5b6_t10.yaml.gz (stand-in for the missing
SHOT forest), annuli=5 bins=6 r_feat=6*mr r_nms=4*mr thr=0.85, draws_remove=false.

Rank 0 prints ONE JSON line.  The oracle (oracle/) is used here only (a) as the parity gate
before timing counts and (b) as the timed `cpu_baseline` -- never on the measured path.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FOREST = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")
A, B = 5, 6
HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def usable_cores():
    """Host cores this process may really use: affinity mask and cgroup CPU quota, not the machine's
    core count (a container on a 256-core host is usually limited to a few of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, 32))


def main():
    # the oracle's OpenMP workers must sleep, not spin, once the parity gate is done: spinning
    # workers would compete with the thread that enqueues the timed steps
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nx", type=int, default=500)
    ap.add_argument("--ny", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--no-parity", action="store_true", help="timing experiments with ablated kernels only")
    ap.add_argument("--detect-only", action="store_true",
                    help="time detectKeypoints only (index prebuilt); reported as extra field anyway")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    local_rank %= torch.cuda.device_count()      # (only matters for the 1-GPU gloo smoke run)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kpl = importlib.import_module("keypoint-learning_amd")
    from tools import synth

    # ---- synthetic view, resident in HBM before the timed region ---------------------------------
    xyz, nrm = synth.make_cloud(args.nx, args.ny, seed=1 + rank)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1001 + rank)
    n = xyz.shape[0]
    thr = float(np.float32(0.85))       # TestDetector parses the threshold as float

    d_xyz = torch.from_numpy(xyz).to(dev)
    d_nrm = torch.from_numpy(nrm).to(dev)
    d_scores = torch.empty(n, dtype=torch.float32, device=dev)
    # keypoint output = one packed buffer [count, idx_0, idx_1, ...]: the engine writes the count and
    # the indices straight into it, and with N > 1 the same buffer is the RCCL all-gather payload
    kp_cap = n
    d_packed = torch.zeros(kp_cap + 1, dtype=torch.int32, device=dev)
    d_cnt, d_kp = d_packed[0:1], d_packed[1:]

    det = kpl.KeypointLearningDetector(device=local_rank)
    mr = det.cloudResolution(xyz)       # kpl_cloud_resolution (input preparation, not timed)
    r_feat = float(np.float32(6.0 * mr))
    r_nms = float(np.float32(4.0 * mr))
    det.setNAnnulus(A)
    det.setNBins(B)
    det.setNonMaxima(True)
    det.setNonMaxRadius(r_nms)
    det.setNonMaximaDrawsRemove(False)
    det.setPredictionThreshold(thr)
    det.setRadiusSearch(r_feat)
    if not det.loadForest(FOREST):
        raise SystemExit("cannot load forest: " + det.lastError())
    det.bindCloudDevice(d_xyz.data_ptr(), 12, d_nrm.data_ptr(), 12, n)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        det.computeDevice(d_scores.data_ptr(), d_kp.data_ptr(), kp_cap, d_cnt.data_ptr(), stream)

    # ---- parity gate (rank 0): keypoint list + scores must equal the oracle's ---------------------
    step()
    if det.syncStatus(stream) == kpl.ERR_RETRY:     # first view of this size: cell tables were grown
        step()
        det.syncStatus(stream)
    torch.cuda.synchronize()
    n_kp = int(d_cnt.item())
    parity = None
    cpu = None
    if rank == 0:
        from oracle import kplo
        from tools import forest_yaml
        fa = forest_yaml.load_forest(FOREST)
        of = kplo.Forest(fa.root, fa.var, fa.thr, fa.left, fa.right, fa.value, fa.var_count)
        ncores = usable_cores()
        o_scores, o_kp = kplo.detect(xyz, nrm, A, B, r_feat, r_nms, thr, of, threads=ncores)
        g_scores = d_scores.cpu().numpy()
        g_kp = d_kp[:n_kp].cpu().numpy()
        same_scores = bool(np.array_equal(g_scores.view(np.uint32), o_scores.view(np.uint32)))
        same_kp = bool(np.array_equal(g_kp, o_kp))
        parity = {"scores_bit_exact": same_scores, "keypoints_identical": same_kp,
                  "n_keypoints": int(len(o_kp))}
        if not (same_scores and same_kp) and not args.no_parity:
            raise SystemExit("PARITY FAILURE vs oracle: %s" % parity)

    # ---- multi-GPU: the one exchange step = gather the keypoint lists ------------------------------
    gather_cap = 32768                     # keypoints per view that travel (the view has ~22 k)
    gathered = [None]
    if world > 1:
        kd = importlib.import_module("keypoint-learning_amd.dist")
        payload = d_packed[:gather_cap + 1]

        def full_step():
            step()
            # the one exchange step of the path: all-gather of the packed keypoint lists (RCCL)
            gathered[0] = kd.gather_keypoints(payload if args.backend == "nccl" else payload.cpu())
    else:
        full_step = step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed settle: let clocks ramp and the host thread pool of the parity gate go to sleep
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.5:
        step()                      # no collective in here: ranks run different iteration counts
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        full_step()
    barrier()
    det.enableTiming(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    t_enq = time.perf_counter()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    timing = det.getTiming()
    det.enableTiming(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        lists = kd.unpack_keypoints(gathered[0])
        assert len(lists) == world and all(len(x) > 0 for x in lists)
        assert np.array_equal(lists[rank].numpy(), d_kp[:len(lists[rank])].cpu().numpy())

    # detect-only timing (index prebuilt: mirrors detectKeypoints without initCompute)
    det.buildIndexDevice(stream)
    torch.cuda.synchronize()
    td0 = time.perf_counter()
    for _ in range(args.steps):
        det.detectDevice(d_scores.data_ptr(), d_kp.data_ptr(), kp_cap, d_cnt.data_ptr(), stream)
    torch.cuda.synchronize()
    detect_only_ms = (time.perf_counter() - td0) * 1e3 / args.steps

    # ---- algorithmic bytes from the engine's own counters -------------------------
    st = det.collectStats(stream)
    b_alg_total = 24 * (st["n_scored"] + st["sum_kf"]) + 16 * st["sum_kn"] + 8 * st["sum_depth"] + 8 * st["n_scored"]
    # share of the dominant kernel (feature + forest): xyz+normal of the point and of each feature
    # neighbor, 8 B per visited forest node, 4 B score out
    b_alg_score = 24 * (st["n_scored"] + st["sum_kf"]) + 8 * st["sum_depth"] + 4 * st["n_scored"]
    score_ms = timing["score_ms"] / max(timing["calls"], 1)
    achieved = b_alg_score / (score_ms * 1e-3) if score_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("score_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    # ---- CPU baseline: the oracle, timed on the host cores, bounded sample --------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        reps, t_cpu = 0, 0.0
        while t_cpu < 8.0 and reps < 8:
            c0 = time.perf_counter()
            kplo.detect(xyz, nrm, A, B, r_feat, r_nms, thr, of, threads=1)
            t_cpu += time.perf_counter() - c0
            reps += 1
        c0 = time.perf_counter()
        kplo.detect(xyz, nrm, A, B, r_feat, r_nms, thr, of, threads=ncores)
        t_all = time.perf_counter() - c0
        cpu = {"value": round(n * reps / t_cpu / 1e6, 4), "unit": "Mpoints/s", "cores": 1, "kind": "port",
               "sample": "%d full passes of the same %d-pt view (grid build + feature + forest + NMS), "
                         "oracle/kpl_oracle.c -O2 -ffp-contract=off, uniform grid not FLANN" % (reps, n),
               "all_cores": {"value": round(n / t_all / 1e6, 4), "cores": ncores}}

    if rank == 0:
        ms = elapsed * 1e3 / args.steps
        out = {
            "metric": "Mpoints/sec scored (feature+forest+NMS), 200k-pt cloud",
            "value": round(n * world * args.steps / elapsed / 1e6, 3),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: single %d-pt synthetic 2.5D view per GPU, 10-tree forest, "
                                   "annuli=5 bins=6 r_feat=6*mr r_nms=4*mr thr=0.85" % n,
                       "points_per_view": n, "views_per_step_per_gpu": 1, "mr": round(mr, 6),
                       "forest": os.path.basename(FOREST), "timed": "index build + detect (compute())",
                       "parallelism": "views sharded, %d rank(s)" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 5), "traffic": traffic,
                         "kernel": "score_kernel (feature + forest)", "kernel_ms": round(score_ms, 5),
                         "alg_bytes_per_launch": int(b_alg_score)},
            "cpu_baseline": cpu,
            "phases_ms": {"index": round(timing["index_ms"] / max(timing["calls"], 1), 5),
                          "score": round(score_ms, 5),
                          "nms_compact": round(timing["nms_ms"] / max(timing["calls"], 1), 5),
                          "detect_only_wall": round(detect_only_ms, 5)},
            "host_enqueue_ms_per_step": round((t_enq - t0) * 1e3 / args.steps, 5),
            "alg_bytes_per_point": round(b_alg_total / max(st["n_scored"], 1), 1),
            "pipeline_alg_GBps": round(b_alg_total / (ms * 1e-3) / 1e9, 2),
            "counters": st,
            "parity": parity,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
