"""Measures the BASELINE.json configs on one MI355X and prints one JSON object per config
(the rows of BASELINE.md section 4).  GPU numbers come from the engine (libkpl, device-resident
inputs, median of timed batches); the CPU columns are the oracle timed in the same process; parity
(scores bit-exact + keypoint lists identical) is checked before any number is reported.

    python tools/run_configs.py [cfg0 cfg1 cfg2 cfg3 cfg4 cfg5]
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import torch  # noqa: E402

from oracle import kplo  # noqa: E402
from tests import helpers  # noqa: E402
from tools import forest_yaml, synth  # noqa: E402

kpl = importlib.import_module("keypoint-learning_amd")
CFG_FOREST = os.path.join(ROOT, "data", "forests", "synth200k_a5b6_t10.yaml.gz")
DEV = torch.device("cuda", 0)


def make_detector(A, B, r, rn, thr, forest, sorted_search=False):
    det = kpl.KeypointLearningDetector()
    det.setNAnnulus(A); det.setNBins(B); det.setNonMaxima(True); det.setNonMaxRadius(rn)
    det.setNonMaximaDrawsRemove(False); det.setPredictionThreshold(thr); det.setRadiusSearch(r)
    det.setSortedSearch(sorted_search)
    if isinstance(forest, str):
        assert det.loadForest(forest), det.lastError()
    else:
        helpers.load_arrays(det, forest)
    return det


def time_gpu(det, xyz, nrm, reps=30, batch=10):
    n = len(xyz)
    dx, dn = torch.from_numpy(np.array(xyz)).to(DEV), torch.from_numpy(np.array(nrm)).to(DEV)
    ds = torch.empty(n, dtype=torch.float32, device=DEV)
    dk = torch.zeros(n + 1, dtype=torch.int32, device=DEV)
    det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
    st = torch.cuda.current_stream().cuda_stream

    def step():
        det.computeDevice(ds.data_ptr(), dk[1:].data_ptr(), n, dk[0:1].data_ptr(), st)
    step()
    while det.syncStatus(st) == kpl.ERR_RETRY:
        step()
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        step()
        torch.cuda.synchronize()
    times = []
    det.enableTiming(True)
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(batch):
            step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / batch)
    tm = det.getTiming()
    det.enableTiming(False)
    cnt = int(dk[0].item())
    return (float(np.median(times)), ds.cpu().numpy(), dk[1:1 + cnt].cpu().numpy(),
            {k: tm[k] / max(tm["calls"], 1) for k in ("index_ms", "score_ms", "feature_ms", "forest_ms", "nms_ms")},
            det.collectStats(st))


def time_cpu(xyz, nrm, A, B, r, rn, thr, of, budget=12.0):
    cores = helpers.usable_cores()
    t0 = time.perf_counter()
    sc, kp = kplo.detect(xyz, nrm, A, B, r, rn, thr, of, threads=cores)
    t_all = time.perf_counter() - t0
    n = len(xyz)
    # 1 thread on a bounded sample: the first m points' worth of work is not separable, so time the
    # whole view only when it fits the budget, else a spatially compact prefix of the storage order
    est = t_all * cores * 0.8
    if est <= budget:
        t0 = time.perf_counter()
        kplo.detect(xyz, nrm, A, B, r, rn, thr, of, threads=1)
        t1 = time.perf_counter() - t0
        sample = "whole view"
        rate1 = n / t1 / 1e6
    else:
        g = kplo.Grid(xyz, r)
        m = max(1000, int(n * budget / est))
        q = g.sorted_indices()[:m].astype(np.int32)
        t0 = time.perf_counter()
        feat = g.features(nrm, A, B, r, q)
        for row in feat[: min(m, 20000)]:
            of.predict_sum(row)
        t1 = time.perf_counter() - t0
        sample = "features of %d points + forest of %d (spatially compact prefix)" % (m, min(m, 20000))
        rate1 = m / t1 / 1e6
    return sc, kp, {"cpu_1thr_Mpts": round(rate1, 4), "cpu_all_Mpts": round(n / t_all / 1e6, 3), "cores": cores,
                    "cpu_sample": sample}


def report(name, xyz, nrm, A, B, rmul_f, rmul_n, thr, forest, fa, extra=None):
    det = make_detector(A, B, 1.0, 1.0, thr, forest)
    mr = det.cloudResolution(xyz)
    r, rn = float(np.float32(rmul_f * mr)), float(np.float32(rmul_n * mr))
    det.setRadiusSearch(r)
    det.setNonMaxRadius(rn)
    of = helpers.oracle_forest(fa)
    o_sc, o_kp, cpu = time_cpu(xyz, nrm, A, B, r, rn, thr, of)
    t, sc, kp, phases, st = time_gpu(det, xyz, nrm)
    ok = bool(np.array_equal(np.asarray(sc).view(np.uint32), o_sc.view(np.uint32)) and np.array_equal(kp, o_kp))
    n = len(xyz)
    b_alg = 24 * (st["n_scored"] + st["sum_kf"]) + 16 * st["sum_kn"] + 8 * st["sum_depth"] + 8 * st["n_scored"]
    b_feat = 24 * (st["n_scored"] + st["sum_kf"])            # gather model, feature kernel / forest kernel
    b_forest = 8 * st["sum_depth"] + 4 * st["n_scored"]
    row = {"config": name, "N": n, "AxB": "%dx%d" % (A, B), "T": fa.ntrees, "nodes": int(fa.nnodes), "r_feat": "%g*mr" % rmul_f,
           "mr": round(mr, 5), "gpu_Mpts": round(n / t / 1e6, 2), "gpu_ms": round(t * 1e3, 4),
           "phases_ms": {k: round(v, 4) for k, v in phases.items()},
           "K_f": round(st["sum_kf"] / max(st["n_scored"], 1), 1), "depth_per_pt": round(st["sum_depth"] / max(st["n_scored"], 1), 1),
           "B_alg_per_pt": round(b_alg / max(st["n_scored"], 1), 1),
           "feature_kernel_alg_GBps": round(b_feat / (phases["feature_ms"] * 1e-3) / 1e9, 1),
           "feature_kernel_frac_of_8TBps": round(b_feat / (phases["feature_ms"] * 1e-3) / 8e12, 4),
           "forest_kernel_alg_GBps": round(b_forest / (phases["forest_ms"] * 1e-3) / 1e9, 1),
           "keypoints": int(len(kp)), "parity": ok}
    row.update(cpu)
    if extra:
        row.update(extra)
    print(json.dumps(row), flush=True)
    assert ok, "PARITY FAILURE in " + name
    return row


def cfg0():
    """The reference's own default operating point (TestDetector with no options, main_test_detector.cpp:62-67, :105-106):
    cheff001, 5 x 10, radiusFeatures 20 / radiusNMS 4 in the cloud's units (~30 / 6 mesh resolutions, K_f ~ 2 900), both
    neighbor orders, checked against the committed oracle outputs (tests/golden/cheff001.npz)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "cheff001.npz"))
    forest = os.path.join(ROOT, "data", "forests", "cheff_a5b10_t10.yaml.gz")
    fa = forest_yaml.load_forest(forest)
    xyz, nrm = z["xyz"], z["nrm"]
    r, rn, thr = float(z["r_feat"]), float(z["r_nms"]), float(z["thr"])
    rows = {}
    cores = helpers.usable_cores()
    walks = [a for a in sys.argv[1:] if a.startswith("walk=")]          # e.g. walk=lanes2 walk=twopass4: forced walks beside the automatic one
    variants = [("canonical", None), ("sorted", None)] + [("canonical", w[5:]) for w in walks]
    for order, forced in variants:
        t0 = time.perf_counter()
        kplo.detect(xyz, nrm, 5, 10, r, rn, thr, helpers.oracle_forest(fa), threads=cores,
                    order=kplo.ORDER_SORTED if order == "sorted" else kplo.ORDER_CANONICAL)
        cpu_all = len(xyz) / (time.perf_counter() - t0) / 1e6
        det = make_detector(5, 10, r, rn, thr, forest, sorted_search=order == "sorted")
        if forced:
            det.setFeatureWalk(kpl.WALK_TWO_PASS if forced.startswith("twopass") else kpl.WALK_LANES, int(forced[-1]))
        mr = det.cloudResolution(xyz)
        t, sc, kp, phases, st = time_gpu(det, xyz, nrm, reps=10, batch=3)
        ok = bool(helpers.same_bits(sc, z["scores_" + order]) and np.array_equal(kp, z["kp_" + order]))
        n = len(xyz)
        b_feat = 24 * (st["n_scored"] + st["sum_kf"])
        wk = det.getFeatureWalk()
        key = order if not forced else forced
        order_name = order if not forced else "canonical order, walk forced to " + forced
        rows[key] = {"config": "cfg0 TestDetector defaults: cheff001, 5x10, r_feat 20 (%.1f mr), r_nms 4, %s order" % (r / mr, order_name),
                       "walk": {"walk": "two-pass" if wk[0] == kpl.WALK_TWO_PASS else "lanes", "lanes_per_point": wk[1], "mean_neighbors_measured": round(wk[2], 1)},
                       "N": n, "T": fa.ntrees, "nodes": int(fa.nnodes), "mr": round(mr, 5), "gpu_Mpts": round(n / t / 1e6, 2),
                       "gpu_ms": round(t * 1e3, 4), "phases_ms": {k: round(v, 4) for k, v in phases.items()},
                       "K_f": round(st["sum_kf"] / max(st["n_scored"], 1), 1), "keypoints": int(len(kp)),
                       "feature_kernel_alg_GBps": round(b_feat / (phases["feature_ms"] * 1e-3) / 1e9, 1),
                       "feature_kernel_frac_of_8TBps": round(b_feat / (phases["feature_ms"] * 1e-3) / 8e12, 4),
                       "cpu_all_Mpts": round(cpu_all, 4), "cores": cores, "parity": ok}
        print(json.dumps(rows[key]), flush=True)
        assert ok, "PARITY FAILURE in cfg0 " + key
    print(json.dumps({"config": "cfg0 sorted / canonical", "compute_ratio": round(rows["sorted"]["gpu_ms"] / rows["canonical"]["gpu_ms"], 3),
                      "feature_kernel_ratio": round(rows["sorted"]["phases_ms"]["feature_ms"] / rows["canonical"]["phases_ms"]["feature_ms"], 3)}),
          flush=True)


def cfg1():
    z = np.load(os.path.join(ROOT, "tests", "golden", "cheff000.npz"))
    fa = forest_yaml.load_forest(CFG_FOREST)
    report("cfg1 cheff000 + cfg forest", z["xyz"], z["nrm"], 5, 6, 6.0, 4.0, float(np.float32(0.85)), CFG_FOREST, fa)


def cfg2():
    xyz, nrm = synth.make_cloud(500, 400, seed=1)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1001)
    fa = forest_yaml.load_forest(CFG_FOREST)
    report("cfg2 synthetic 200k", xyz, nrm, 5, 6, 6.0, 4.0, float(np.float32(0.85)), CFG_FOREST, fa)


def cfg3():
    """64 views of ~63 k points on ONE GPU (the per-GPU share of the 8-GPU config is 8 views): batches
    of 8 views (kpl_compute_batch_device), two batches in flight on two HIP streams, 16 distinct views
    resident, 4 rounds = 64 views."""
    fa = forest_yaml.load_forest(CFG_FOREST)
    views = []
    for k in range(16):
        xyz, nrm = synth.make_cloud(252, 250, seed=100 + k)
        views.append(synth.shuffle_cloud(xyz, nrm, 1100 + k))
    dets, bufs = [], []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for xyz, nrm in views:
        det = make_detector(5, 6, 1.0, 1.0, float(np.float32(0.85)), CFG_FOREST)
        mr = det.cloudResolution(xyz)
        det.setRadiusSearch(float(np.float32(6 * mr)))
        det.setNonMaxRadius(float(np.float32(4 * mr)))
        n = len(xyz)
        dx, dn = torch.from_numpy(np.array(xyz)).to(DEV), torch.from_numpy(np.array(nrm)).to(DEV)
        ds = torch.empty(n, dtype=torch.float32, device=DEV)
        dk = torch.zeros(n + 1, dtype=torch.int32, device=DEV)
        det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
        dets.append(det); bufs.append((dx, dn, ds, dk))

    def sweep():
        for g in range(2):
            sl = slice(8 * g, 8 * g + 8)
            kpl.compute_batch_device(dets[sl], [b[2].data_ptr() for b in bufs[sl]], [b[3][1:].data_ptr() for b in bufs[sl]],
                                     [len(b[2]) for b in bufs[sl]], [b[3][0:1].data_ptr() for b in bufs[sl]],
                                     streams[g].cuda_stream)
    torch.cuda.synchronize()            # (the buffers above were filled on the default stream; the sweeps run on others)
    sweep()
    torch.cuda.synchronize()
    while kpl.ERR_RETRY in [det.syncStatus(None) for det in dets]:   # every detector, no short circuit
        sweep()
        torch.cuda.synchronize()
    sweep(); torch.cuda.synchronize()
    ok = True
    for (xyz, nrm), det, (dx, dn, ds, dk) in zip(views, dets, bufs):
        p = det._p
        o_sc, o_kp = kplo.detect(xyz, nrm, 5, 6, p.radius_search, p.non_max_radius, p.prediction_th,
                                 helpers.oracle_forest(fa), threads=helpers.usable_cores())
        cnt = int(dk[0].item())
        ok &= bool(np.array_equal(ds.cpu().numpy().view(np.uint32), o_sc.view(np.uint32)) and
                   np.array_equal(dk[1:1 + cnt].cpu().numpy(), o_kp))
    times = []
    for _ in range(20):
        t0 = time.perf_counter()
        for _ in range(4):          # 4 rounds x 2 batches x 8 views = the 64 views of the config
            sweep()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    npts = 4 * sum(len(v[0]) for v in views)
    print(json.dumps({"config": "cfg3 64 views x 63k on ONE GPU, batches of 8, 2 in flight", "points": npts,
                      "makespan_ms": round(t * 1e3, 3), "gpu_Mpts": round(npts / t / 1e6, 2), "parity": ok}), flush=True)
    assert ok


def cfg4():
    xyz, nrm = synth.make_cloud(707, 707, seed=4)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1004)
    fa = forest_yaml.load_forest(CFG_FOREST)
    for rmul in (4.0, 6.0, 8.0, 10.0):
        report("cfg4 dense 500k r=%g*mr" % rmul, xyz, nrm, 5, 6, rmul, 4.0, float(np.float32(0.85)), CFG_FOREST, fa)


def cfg5():
    xyz, nrm = synth.make_cloud(500, 500, seed=5, overlap_layers=4)
    xyz, nrm = synth.shuffle_cloud(xyz, nrm, 1005)
    A, B = 8, 10
    det = kpl.KeypointLearningDetector()
    mr = det.cloudResolution(xyz)
    r = float(np.float32(6 * mr))
    g = kplo.Grid(xyz, r)
    feat = g.features(nrm, A, B, r, g.sorted_indices()[::97].astype(np.int32))
    t0 = time.perf_counter()
    fa = synth.random_forest(A * B, ntrees=100, max_depth=28, seed=3, target_nodes_per_tree=20000, feat=feat)
    report("cfg5 fused 1M, 100 deep trees", xyz, nrm, A, B, 6.0, 4.0, float(np.float32(0.3)), fa, fa,
           {"forest_build_s": round(time.perf_counter() - t0, 1)})


def cfg0b():
    """The reference's default operating point as a BATCH: 8 cheff views (cheff000 / 001 / 002, the three clouds of
    data/point_cloud_test, cycled) through kpl_compute_batch_device at radiusFeatures 20 / radiusNMS 4, 5 x 10 -- what
    DetectViews does with the views of a dataset.  Views checked against the oracle's scores of the same view."""
    forest = os.path.join(ROOT, "data", "forests", "cheff_a5b10_t10.yaml.gz")
    fa = forest_yaml.load_forest(forest)
    z1 = np.load(os.path.join(ROOT, "tests", "golden", "cheff001.npz"))
    r, rn, thr = float(z1["r_feat"]), float(z1["r_nms"]), float(z1["thr"])
    clouds = [np.load(os.path.join(ROOT, "tests", "golden", "cheff00%d.npz" % k)) for k in (1, 0, 2)]
    of = helpers.oracle_forest(fa)
    expect = [kplo.detect(c["xyz"], c["nrm"], 5, 10, r, rn, thr, of, threads=helpers.usable_cores()) for c in clouds]
    for order in ("canonical", "sorted"):
        if order == "sorted":
            expect = [kplo.detect(c["xyz"], c["nrm"], 5, 10, r, rn, thr, of, threads=helpers.usable_cores(), order=kplo.ORDER_SORTED)
                      for c in clouds]
        dets, bufs = [], []
        for k in range(8):
            c = clouds[k % 3]
            det = make_detector(5, 10, r, rn, thr, forest, sorted_search=order == "sorted")
            n = len(c["xyz"])
            dx, dn = torch.from_numpy(np.array(c["xyz"])).to(DEV), torch.from_numpy(np.array(c["nrm"])).to(DEV)
            ds = torch.empty(n, dtype=torch.float32, device=DEV)
            dk = torch.zeros(n + 1, dtype=torch.int32, device=DEV)
            det.bindCloudDevice(dx.data_ptr(), 12, dn.data_ptr(), 12, n)
            dets.append(det); bufs.append((dx, dn, ds, dk))
        stream = torch.cuda.Stream()

        def sweep():
            kpl.compute_batch_device(dets, [b[2].data_ptr() for b in bufs], [b[3][1:].data_ptr() for b in bufs],
                                     [len(b[2]) for b in bufs], [b[3][0:1].data_ptr() for b in bufs], stream.cuda_stream)
        torch.cuda.synchronize()
        for _ in range(6):          # tables grow, then the handles learn their neighborhood size
            sweep()
            torch.cuda.synchronize()
            if kpl.ERR_RETRY not in [det.syncStatus(None) for det in dets]:
                pass
        sweep(); torch.cuda.synchronize()
        assert all(det.syncStatus(None) == kpl.OK for det in dets)
        ok = True
        for k, (det, (dx, dn, ds, dk)) in enumerate(zip(dets, bufs)):
            o_sc, o_kp = expect[k % 3]
            cnt = int(dk[0].item())
            ok &= bool(helpers.same_bits(ds.cpu().numpy(), o_sc) and np.array_equal(dk[1:1 + cnt].cpu().numpy(), o_kp))
        dets[0].enableTiming(True)
        times = []
        for _ in range(12):
            t0 = time.perf_counter()
            sweep(); sweep()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / 2)
        tm = dets[0].getTiming()
        dets[0].enableTiming(False)
        t = float(np.median(times))
        npts = sum(len(b[2]) for b in bufs)
        wk = dets[0].getFeatureWalk()
        print(json.dumps({"config": "cfg0b 8 cheff views in one batch at the reference's defaults (5x10, r_feat 20, r_nms 4), %s order" % order,
                          "points": npts, "ms_per_batch": round(t * 1e3, 4), "gpu_Mpts": round(npts / t / 1e6, 2),
                          "walk": {"walk": "two-pass" if wk[0] == kpl.WALK_TWO_PASS else "lanes", "lanes_per_point": wk[1], "mean_neighbors_measured": round(wk[2], 1)},
                          "phases_ms": {k: round(tm[k] / max(tm["calls"], 1), 4) for k in ("index_ms", "feature_ms", "forest_ms", "nms_ms")},
                          "parity": ok}), flush=True)
        assert ok, "PARITY FAILURE in cfg0b " + order


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("walk=")] or ["cfg0", "cfg1", "cfg2", "cfg3", "cfg4", "cfg5"]
    for name in which:
        globals()[name]()
